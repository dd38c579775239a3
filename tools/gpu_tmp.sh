cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5final; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 3000 python -m pytest tests -m gpu -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -4 $O/pytest.log
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err
timeout 600 python bench.py --steps 20 --warmup 5 --overlap 0 --no-secondary --no-cpu-baseline > $O/bench_k20_one_queue.json 2>/dev/null
timeout 600 python bench.py --overlap 0 --no-secondary --no-cpu-baseline > $O/bench_n1_one_queue.json 2>/dev/null
DIF_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --no-secondary > $O/bench_rccl_1rank.json 2> /dev/null
DIF_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --no-secondary --rccl-before-clock 1 > $O/bench_rccl_1rank_before_clock.json 2> /dev/null
timeout 1200 python tools/soak_overlap.py 300 60 2>&1 | tail -1 > $O/soak_overlap.txt; cat $O/soak_overlap.txt
timeout 600 python tools/determinism_stress.py 60 2>&1 | tail -1 > $O/determinism.txt; cat $O/determinism.txt
python - $O <<'PY'
import json,sys
for n in ("bench_n1","bench_k20","bench_k20_one_queue","bench_n1_one_queue","bench_rccl_1rank","bench_rccl_1rank_before_clock"):
    d=json.loads(open(f"{sys.argv[1]}/{n}.json").read().strip().splitlines()[-1]); c=d["config"]; r=d["roofline"]
    print(n, d["value"], d["ms_per_step"], r["frac"], r.get("pmc_mfma_busy_frac"), d.get("rccl_ranks"), c.get("rccl_before_clock"), c.get("frames_per_s_with_the_mesh_half_on_a_third_queue_and_the_host_two_frames_ahead"), (d.get("cpu_baseline") or {}).get("value"))
PY
