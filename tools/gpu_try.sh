#!/bin/bash
# a quick look after a kernel change: the map / marching-cubes / stream / tiling tests, two bench lines, steady-state and whole-run kernel stats of the changed kernels
tag=${1:-q}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_map.py tests/test_gpu_mc_exhaustive.py tests/test_gpu_stream.py tests/test_gpu_parallel.py "tests/test_gpu_fuzz.py::test_fuzz_marching_cubes" "tests/test_gpu_fuzz.py::test_fuzz_integrate_extract_query[0]" -m gpu -q -x 2>&1 | tail -6
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $out/bench_$i.json
python - $out/bench_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["other_ms_per_frame"])
PY
done
timeout 400 rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python bench.py --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/rocpd_stats.py $(find $out/trace -name "*.db" | head -1) --after-nth k_prune_mark 160 --frames 50 | grep -i "onepass\|finish\|sum of\|dirty\|Occ\|Alloc"
python tools/rocpd_stats.py $(find $out/trace -name "*.db" | head -1) --after-nth k_prune_mark 14 --frames 200 | grep -i "onepass\|finish\|sum of\|dirty\|Occ\|Alloc"
rm -rf $out/trace
