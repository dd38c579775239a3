#!/bin/bash
# the one-pass marching cubes of a stream frame with and without the XCD-aware dealing of groups: launch time (HIP events) and counter traffic.
# Build the variant HERE first (ab_old/ travels with the snapshot): hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared
#   -DNDEBUG -DMC_XCD_RUN=1 -DDIF_BUILD_ID='"sw"' di_fusion_amd/csrc/difusion.hip -o ab_old/libdif_run1.so      (MC_XCD_RUN=1: group = workgroup index)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/swz; mkdir -p $out
for v in run1 main run1 main; do
  if [ $v = main ]; then unset DIF_LIB; else export DIF_LIB=$GRAFT_REPO_ROOT/ab_old/libdif_$v.so; fi
  for args in "" "--steps 20 --warmup 5" "--streams-per-gpu 8"; do
    timeout 300 python bench.py --no-cpu-baseline --no-secondary --d2h none $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '$args', d['value'], d['roofline'].get('other_ms_per_frame'))"
  done
  n=$((n+1)); [ $n -le 2 ] && for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $c --output-format csv -d $out/$v$c -o p -- python bench.py --no-cpu-baseline --no-secondary --steps 40 --warmup 4 > $out/$v$c.log 2>&1
  done
  [ $n -le 2 ] && python tools/pmc_summary.py $(find $out/${v}FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $out/${v}WRITE_SIZE -name "*counter_collection.csv" | head -1) $out/pmc_$v.json > /dev/null 2>&1
  [ $n -le 2 ] && python -c "
import json; d=json.load(open('$out/pmc_$v.json'))['kernels']
print('$v', {k:v for k,v in d.items() if 'marching' in k})"
  rm -rf $out/${v}FETCH_SIZE $out/${v}WRITE_SIZE
done
