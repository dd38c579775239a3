#!/bin/bash
# S streams per GPU (dif_integrate_frames / dif_extract_streams): the group tests, then the bench with its by-streams legs and explicit --streams-per-gpu runs
# usage (via gpurun): bash tools/gpu_streams.sh <tag>
tag=${1:-s}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_stream.py -m gpu -q -x 2>&1 | tail -15 | cut -c1-400
timeout 600 python bench.py --no-cpu-baseline > $out/bench_200.json 2> $out/bench_200.err; tail -3 $out/bench_200.err
python - $out/bench_200.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("single", d["value"], d["ms_per_step"], d["config"].get("frames_per_s_with_S_streams_per_gpu"))
for S,b in (r.get("by_streams") or {}).items():
    print(S, json.dumps({k:(b[k] if k!="per_kernel" else {n:(v["ms_per_launch"], v["frac"]) for n,v in b[k].items()}) for k in b if k in ("per_kernel","other_ms_per_frame","error")}) if b else None)
PY
for S in 1 4; do
timeout 300 python bench.py --no-cpu-baseline --no-secondary --streams-per-gpu $S > $out/bench_s$S.json 2> $out/bench_s$S.err; tail -2 $out/bench_s$S.err; cut -c1-300 $out/bench_s$S.json
done
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $out/bench_k20.json 2> $out/bench_k20.err; tail -2 $out/bench_k20.err; cut -c1-200 $out/bench_k20.json
