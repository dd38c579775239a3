# usage (GPU box, via gpurun): bash tools/refresh_profiles.sh <tag>      -> gpurun_out/<tag>/ ; tools/install_profiles.sh copies the summaries to profiles/
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-final1}; mkdir -p $O
B="python bench.py"
# ---- the default invocation (200 steps) ----
timeout 600 $B > $O/bench_n1.json 2> $O/bench_n1.err
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- $B --no-cpu-baseline --no-secondary > $O/trace_bench.log 2>&1
python tools/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) --after-nth k_prune_mark 14 --frames 200 > $O/kernel_stats.md 2>&1
python tools/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) --after-nth k_prune_mark 160 --frames 50 > $O/kernel_stats_steady.md 2>&1
python tools/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) --timeline k_prune_mark 150 > $O/timeline_overlap.txt 2>&1
rm -rf $O/trace
# ---- the same stream on ONE queue (--overlap 0): what the second queue buys ----
timeout 600 $B --no-cpu-baseline --no-secondary --overlap 0 > $O/bench_n1_one_queue.json 2> $O/bench_n1_one_queue.err
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- $B --no-cpu-baseline --no-secondary --overlap 0 > $O/trace_bench_ov0.log 2>&1
python tools/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) --after-nth k_prune_mark 160 --frames 50 > $O/kernel_stats_steady_one_queue.md 2>&1
python tools/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) --timeline k_prune_mark 150 > $O/timeline_direct.txt 2>&1
rm -rf $O/trace
timeout 300 $B --no-cpu-baseline --no-secondary --overlap 0 --steps 20 --warmup 5 > $O/bench_k20_one_queue.json 2> $O/bench_k20_one_queue.err
# ---- the DRIVER's invocation: --steps 20 --warmup 5 (map-building transient) ----
K20="--steps 20 --warmup 5 --no-cpu-baseline"
timeout 300 $B $K20 > $O/bench_k20.json 2> $O/bench_k20.err
# (the rocprofv3 passes of this invocation run WITHOUT the secondary legs — the S-streams-per-GPU rates that follow the timed region since round 4:
# their streams' first frames launch the same kernels and would land in "the last 20 dispatches"; the timed region itself is identical)
K20="$K20 --no-secondary"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace20 -o bench -- $B $K20 > $O/trace20.log 2>&1
python tools/rocpd_stats.py $(find $O/trace20 -name "*.db" | head -1) --after-nth k_prune_mark 9 --frames 20 > $O/kernel_stats_k20.md 2>&1
rm -rf $O/trace20
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma20 -o m -- $B $K20 > $O/pmc_mfma20.log 2>&1
python tools/pmc_mfma.py $(find $O/pmc_mfma20 -name "*counter_collection.csv" | head -1) $O/pmc_mfma_k20.json --last 20 > $O/pmc_mfma20_summary.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch20 -o f -- $B $K20 > $O/pmc_fetch20.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write20 -o w -- $B $K20 > $O/pmc_write20.log 2>&1
python tools/pmc_summary.py $(find $O/pmc_fetch20 -name "*counter_collection.csv" | head -1) $(find $O/pmc_write20 -name "*counter_collection.csv" | head -1) $O/pmc_hbm_k20.json --last 20 > $O/pmc_summary20.log 2>&1
rm -rf $O/pmc_mfma20 $O/pmc_fetch20 $O/pmc_write20
# ---- PMC passes of the 200-step stream ----
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- $B --no-cpu-baseline --no-secondary --steps 40 --warmup 4 > $O/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- $B --no-cpu-baseline --no-secondary --steps 40 --warmup 4 > $O/pmc_write.log 2>&1
python tools/pmc_summary.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) $O/pmc_hbm.json > $O/pmc_summary.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o m -- $B --no-cpu-baseline --no-secondary > $O/pmc_mfma.log 2>&1
python tools/pmc_mfma.py $(find $O/pmc_mfma -name "*counter_collection.csv" | head -1) $O/pmc_mfma_stream.json > $O/pmc_mfma_summary.log 2>&1
rm -rf $O/pmc_mfma $O/pmc_fetch $O/pmc_write
# ---- S streams per launch (bench.py --streams-per-gpu S): kernel stats, PMC MFMA busy and HBM bytes, 200-step and the driver's K = 20 shape ----
for S in 2 4 8; do
  BS="$B --no-cpu-baseline --no-secondary --streams-per-gpu $S"
  timeout 300 $BS > $O/bench_s$S.json 2> $O/bench_s$S.err
  timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_ms -o m -- $BS > $O/pmc_mfma_s$S.log 2>&1
  python tools/pmc_mfma.py $(find $O/pmc_ms -name "*counter_collection.csv" | head -1) $O/pmc_mfma_s$S.json > $O/pmc_mfma_s${S}_summary.log 2>&1
  rm -rf $O/pmc_ms
  timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_ms -o m -- $BS --steps 20 --warmup 5 > $O/pmc_mfma_s${S}_k20.log 2>&1
  python tools/pmc_mfma.py $(find $O/pmc_ms -name "*counter_collection.csv" | head -1) $O/pmc_mfma_s${S}_k20.json --last 20 > $O/pmc_mfma_s${S}_k20_summary.log 2>&1
  rm -rf $O/pmc_ms
done
for S in 4; do
  BS="$B --no-cpu-baseline --no-secondary --streams-per-gpu $S"
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace_s -o bench -- $BS > $O/trace_s$S.log 2>&1
  python tools/rocpd_stats.py $(find $O/trace_s -name "*.db" | head -1) --after-nth k_prune_mark 160 --frames 50 > $O/kernel_stats_s${S}_steady.md 2>&1
  python tools/rocpd_stats.py $(find $O/trace_s -name "*.db" | head -1) --after-nth k_prune_mark 9 --frames 200 > $O/kernel_stats_s$S.md 2>&1
  python tools/rocpd_stats.py $(find $O/trace_s -name "*.db" | head -1) --timeline k_prune_mark 150 > $O/timeline_s$S.txt 2>&1
  rm -rf $O/trace_s
  timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fs -o f -- $BS --steps 40 --warmup 4 > $O/pmc_fetch_s$S.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_ws -o w -- $BS --steps 40 --warmup 4 > $O/pmc_write_s$S.log 2>&1
  python tools/pmc_summary.py $(find $O/pmc_fs -name "*counter_collection.csv" | head -1) $(find $O/pmc_ws -name "*counter_collection.csv" | head -1) $O/pmc_hbm_s$S.json > $O/pmc_summary_s$S.log 2>&1
  rm -rf $O/pmc_fs $O/pmc_ws
done
bash tools/gpu_frows_prof.sh $(basename $O) > $O/frows.log 2>&1
# ---- C5: the 1280x960 stream on one GPU, and slab 4 of 8 exchanging halos with itself ----
timeout 600 $B --mode tiled --no-cpu-baseline --steps 100 > $O/bench_tiled_n1.json 2> $O/bench_tiled_n1.err
timeout 600 $B --mode tiled --loopback 8 --no-cpu-baseline --steps 100 > $O/bench_tiled_loopback8_delta.json 2> $O/bench_tiled_loopback8_delta.err
timeout 600 $B --mode tiled --loopback 8 --halo full --no-cpu-baseline --steps 100 > $O/bench_tiled_loopback8_full.json 2> $O/bench_tiled_loopback8_full.err
timeout 600 $B --mode tiled --loopback 2 --no-cpu-baseline --steps 100 > $O/bench_tiled_loopback2_delta.json 2> $O/bench_tiled_loopback2_delta.err
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace_lb8 -o bench -- $B --mode tiled --loopback 8 --no-cpu-baseline --steps 100 > $O/trace_lb8.log 2>&1
python tools/rocpd_stats.py $(find $O/trace_lb8 -name "*.db" | head -1) --after-nth k_prune_mark 14 --frames 100 > $O/kernel_stats_tiled_loopback8.md 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace_t1 -o bench -- $B --mode tiled --no-cpu-baseline --steps 100 > $O/trace_t1.log 2>&1
python tools/rocpd_stats.py $(find $O/trace_t1 -name "*.db" | head -1) --after-nth k_prune_mark 14 --frames 100 > $O/kernel_stats_tiled_n1.md 2>&1
rm -rf $O/trace_lb8 $O/trace_t1
# ---- the rest ----
for c in c1 c2; do timeout 300 $B --config $c --no-cpu-baseline --steps 50 > $O/bench_$c.json 2> $O/bench_$c.err; done
DIF_FORCE_DIST=1 timeout 300 $B --no-cpu-baseline --no-secondary > $O/bench_rccl_1rank.json 2> $O/bench_rccl_1rank.err
DIF_FORCE_DIST=1 timeout 300 $B --no-cpu-baseline --no-secondary --rccl-before-clock 1 > $O/bench_rccl_1rank_before_clock.json 2> $O/bench_rccl_1rank_before_clock.err
timeout 900 python tools/stress_full_occupancy.py --n 128 --reps 2 > $O/stress_full.json 2> $O/stress_full.err
timeout 900 python tools/stress_integrate.py > $O/stress_integrate.json 2> $O/stress_integrate.err
timeout 300 python tools/bench_cloud.py --cpu-sample 20000 > $O/bench_cloud.json 2> $O/bench_cloud.err
timeout 300 python tools/bench_query.py > $O/bench_query.json 2> $O/bench_query.err
timeout 300 python tools/sweep_decode.py > $O/sweep_decode.jsonl 2> $O/sweep_decode.err
ls -la $O | wc -l
