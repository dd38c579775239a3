#!/bin/bash
# kernel trace + PMC (MFMA busy) of the stream group at S streams per GPU, steady state.  usage (via gpurun): bash tools/gpu_streams_prof.sh <tag> "<S list>"
tag=${1:-sp}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for S in ${2:-4}; do
B="python bench.py --no-cpu-baseline --no-secondary --streams-per-gpu $S"
timeout 400 rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- $B > $out/trace_s$S.log 2>&1
python tools/rocpd_stats.py $(find $out/trace -name "*.db" | head -1) --after-nth k_prune_mark 160 --frames 50 > $out/kernel_stats_s${S}_steady.md 2>&1
python tools/rocpd_stats.py $(find $out/trace -name "*.db" | head -1) --after-nth k_prune_mark 9 --frames 200 > $out/kernel_stats_s${S}.md 2>&1
python tools/rocpd_stats.py $(find $out/trace -name "*.db" | head -1) --timeline k_prune_mark 150 > $out/timeline_s$S.txt 2>&1
rm -rf $out/trace
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc -o m -- $B > $out/pmc_s$S.log 2>&1
python tools/pmc_mfma.py $(find $out/pmc -name "*counter_collection.csv" | head -1) $out/pmc_mfma_s$S.json > $out/pmc_mfma_s${S}_summary.log 2>&1
rm -rf $out/pmc
cat $out/kernel_stats_s${S}_steady.md; cat $out/pmc_mfma_s${S}_summary.log
done
