set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; mkdir -p $O
B="python bench.py"
K20="--steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace20 -o bench -- $B $K20 > $O/trace20.log 2>&1
python tools/rocpd_stats.py $(find $O/trace20 -name "*.db" | head -1) --after-nth k_prune_mark 9 --frames 20 > $O/kernel_stats_k20.md 2>&1
rm -rf $O/trace20
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma20 -o m -- $B $K20 > $O/pmc_mfma20.log 2>&1
python tools/pmc_mfma.py $(find $O/pmc_mfma20 -name "*counter_collection.csv" | head -1) $O/pmc_mfma_k20.json --last 20 > $O/pmc_mfma20_summary.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch20 -o f -- $B $K20 > $O/pmc_fetch20.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write20 -o w -- $B $K20 > $O/pmc_write20.log 2>&1
python tools/pmc_summary.py $(find $O/pmc_fetch20 -name "*counter_collection.csv" | head -1) $(find $O/pmc_write20 -name "*counter_collection.csv" | head -1) $O/pmc_hbm_k20.json --last 20 > $O/pmc_summary20.log 2>&1
rm -rf $O/pmc_mfma20 $O/pmc_fetch20 $O/pmc_write20
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_k20_full_$i.json 2> $O/bench_k20_full_$i.err; done
for i in 1 2 3; do python tools/bench_query.py > $O/bench_query_$i.json 2>/dev/null; done
cat $O/pmc_mfma20_summary.log; head -20 $O/kernel_stats_k20.md
