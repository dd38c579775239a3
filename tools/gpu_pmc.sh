#!/bin/bash
tag=${1:-pmc}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_WAVES SQ_IFETCH SQ_IFETCH_LEVEL GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --output-format csv -d $out/p$i -o p -- python bench.py --no-cpu-baseline --no-secondary --steps 60 --warmup 100 > $out/p$i.log 2>&1
done
python tools/pmc_any.py $(find $out -name "*counter_collection.csv") | tee $out/pmc_any.txt
rm -rf $out/p1 $out/p2 $out/p3
