#!/bin/bash
# A/B of library builds on the stream bench: steady-state per-kernel averages (rocprofv3 kernel trace) and frames/s.
# usage: bash tools/gpu_ab_kernels.sh "<kernel name pattern>" <name> ...    (ab_old/libdif_<name>.so; "main" = the in-tree library)
pat=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/abk; mkdir -p $out
for v in "$@"; do
  if [ $v = main ]; then unset DIF_LIB; else export DIF_LIB=$GRAFT_REPO_ROOT/ab_old/libdif_$v.so; fi
  for i in 1 2; do
    timeout 300 python bench.py --no-cpu-baseline --no-secondary --d2h none 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
  done
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --d2h none --streams-per-gpu 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v S=8', d['value'], d['ms_per_step'])"
  rm -rf $out/trace
  timeout 400 rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python bench.py --no-cpu-baseline --no-secondary > /dev/null 2>&1
  python tools/rocpd_stats.py $(find $out/trace -name "*.db" | head -1) --after-nth k_prune_mark 160 --frames 50 | grep -E "$pat|sum of kernel" | sed "s/^/$v /"
  rm -rf $out/trace
done
