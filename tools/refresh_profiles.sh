set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-final1}; mkdir -p $O
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python bench.py --no-cpu-baseline --no-secondary > $O/trace_bench.log 2>&1
python tools/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) --after-nth k_prune_mark 14 --frames 200 > $O/kernel_stats.md 2>&1
python tools/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) --timeline k_prune_mark 150 > $O/timeline_direct.txt 2>&1
timeout 400 rocprofv3 --kernel-trace -d $O/trace_g -o bench -- python bench.py --graph 1 --no-cpu-baseline --no-secondary --steps 100 > $O/trace_graph.log 2>&1
python tools/rocpd_stats.py $(find $O/trace_g -name "*.db" | head -1) --timeline k_prune_mark 90 > $O/timeline_graph.txt 2>&1
rm -rf $O/trace_g
timeout 300 python bench.py --graph 1 --no-cpu-baseline > $O/bench_graph.json 2> $O/bench_graph.err
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python bench.py --no-cpu-baseline --no-secondary --steps 40 --warmup 4 --graph 0 > $O/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python bench.py --no-cpu-baseline --no-secondary --steps 40 --warmup 4 --graph 0 > $O/pmc_write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o m -- python bench.py --no-cpu-baseline --no-secondary --steps 200 --warmup 10 --graph 0 > $O/pmc_mfma.log 2>&1
python tools/pmc_mfma.py $(find $O/pmc_mfma -name "*counter_collection.csv" | head -1) $O/pmc_mfma_stream.json > $O/pmc_mfma_summary.log 2>&1
rm -rf $O/pmc_mfma
python tools/pmc_summary.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) $O/pmc_hbm.json > $O/pmc_summary.log 2>&1
timeout 900 python tools/stress_full_occupancy.py --n 128 --reps 2 > $O/stress_full.json 2> $O/stress_full.err
timeout 900 python tools/stress_integrate.py > $O/stress_integrate.json 2> $O/stress_integrate.err
timeout 300 python tools/bench_cloud.py --cpu-sample 20000 > $O/bench_cloud.json 2> $O/bench_cloud.err
for c in c1 c2; do timeout 300 python bench.py --config $c --no-cpu-baseline --steps 50 > $O/bench_$c.json 2> $O/bench_$c.err; done
timeout 600 python bench.py --mode tiled --no-cpu-baseline --steps 100 > $O/bench_tiled_n1.json 2> $O/bench_tiled_n1.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_k20.json 2> $O/bench_k20.err
timeout 300 python tools/bench_query.py > $O/bench_query.json 2> $O/bench_query.err
rm -rf $O/pmc_fetch $O/pmc_write
find $O/trace -name "*.db" -size +20M -delete
ls -la $O
timeout 300 python bench.py --batch 5 --no-cpu-baseline --no-secondary > $O/bench_batch5.json 2> $O/bench_batch5.err
DIF_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --steps 50 --no-secondary > $O/bench_rccl_1rank.json 2> $O/bench_rccl_1rank.err
DIF_DECODER_PIPE=f32 timeout 300 python bench.py --no-cpu-baseline --no-secondary > $O/bench_f32pipe.json 2> $O/bench_f32pipe.err
ls -la $O | wc -l
