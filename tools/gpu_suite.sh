#!/bin/bash
# the whole -m gpu suite, then the bench (default + the driver's invocation).  usage (via gpurun): bash tools/gpu_suite.sh <tag> [pytest args]
tag=${1:-suite}; shift; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 3000 python -m pytest tests -m gpu -q --durations=15 "$@" > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -40 $out/pytest.log | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline > $out/bench_200.json 2> $out/bench_200.err; tail -3 $out/bench_200.err
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $out/bench_k20.json 2> $out/bench_k20.err; tail -2 $out/bench_k20.err
python - $out/bench_200.json $out/bench_k20.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f, d["value"], d["ms_per_step"], d["config"].get("frames_per_s_with_S_streams_per_gpu"), d["config"].get("frames_per_s_with_mesh_left_in_hbm"))
PY
