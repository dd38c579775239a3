#!/bin/bash
# round 5, first visit: runtime questions (cross-queue hand-off, the export copy), the new long-sequence parity tests, the whole suite, a baseline bench
tag=${1:-r5a}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 tools/micro/handoff > $out/handoff.txt 2>&1; echo "handoff rc=$?"
GPU_FORCE_BLIT_COPY_SIZE=0 timeout 300 tools/micro/handoff > $out/handoff_blit0.txt 2>&1; echo "handoff blit0 rc=$?"
GPU_MAX_HW_QUEUES=8 timeout 300 tools/micro/handoff > $out/handoff_q8.txt 2>&1; echo "handoff q8 rc=$?"
for v in default blit0; do
  [ $v = blit0 ] && export GPU_FORCE_BLIT_COPY_SIZE=0
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $GRAFT_REPO_ROOT/$out/prof_handoff_$v -o h -- $GRAFT_REPO_ROOT/tools/micro/handoff > /dev/null 2>&1)
  unset GPU_FORCE_BLIT_COPY_SIZE
  f=$(ls $out/prof_handoff_$v/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && { echo "== kernel stats ($v)"; head -8 $f | cut -c1-200; } > $out/handoff_kernels_$v.txt
  f=$(ls $out/prof_handoff_$v/*memory_copy_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && { echo "== memory copy stats ($v)"; head -8 $f | cut -c1-200; } >> $out/handoff_kernels_$v.txt
  cat $out/handoff_kernels_$v.txt
  rm -rf $out/prof_handoff_$v
done
head -60 $out/handoff.txt
timeout 2400 python -m pytest tests/test_gpu_long.py -m gpu -q -s -x > $out/pytest_long.log 2>&1; echo "long rc=$?"; tail -25 $out/pytest_long.log | cut -c1-400
timeout 3000 python -m pytest tests -m gpu -q --durations=12 --deselect tests/test_gpu_long.py > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -30 $out/pytest.log | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $out/bench_k20.json 2> $out/bench_k20.err; tail -2 $out/bench_k20.err
timeout 600 python bench.py --no-cpu-baseline --no-secondary > $out/bench_200.json 2> $out/bench_200.err; tail -3 $out/bench_200.err
python - $out/bench_200.json $out/bench_k20.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f, d["value"], d["ms_per_step"], r["frac"], {k: (v["ms_per_launch"], v["frac"]) for k, v in r["per_kernel"].items()})
    except Exception as e:
        print(f, "unreadable", e)
PY
