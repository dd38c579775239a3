#!/bin/bash
# rocprofv3 kernel trace + PMC (MFMA busy, FETCH_SIZE, WRITE_SIZE: separate passes) of the section-8f kernels: get_sdf (+gradient), latent optimisation,
# cloud ops, image front end.  usage (via gpurun): bash tools/gpu_frows_prof.sh <tag>
tag=${1:-fr}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for name in query cloud frows; do
  cmd="python tools/bench_$name.py"
  timeout 300 $cmd > $out/bench_$name.json 2> $out/bench_$name.err; tail -2 $out/bench_$name.err
  timeout 400 rocprofv3 --kernel-trace --stats -d $out/t_$name -o t -- $cmd > $out/trace_$name.log 2>&1
  timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/m_$name -o m -- $cmd > $out/pmc_m_$name.log 2>&1
  timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f_$name -o f -- $cmd > $out/pmc_f_$name.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/w_$name -o w -- $cmd > $out/pmc_w_$name.log 2>&1
  python tools/frows_summary.py $(find $out/t_$name -name "*.db" | head -1) --mfma $(find $out/m_$name -name "*counter_collection.csv" | head -1) \
     --fetch $(find $out/f_$name -name "*counter_collection.csv" | head -1) --write $(find $out/w_$name -name "*counter_collection.csv" | head -1) \
     --only k_decode_grad,k_decode_x6,k_optim,k_cloud,k_depth_frontend,k_pbf,k_query,k_sdf_hg,QueryFunctor,OptimGather,OptimUnique,CloudStart,BoxRank > $out/kernel_stats_$name.md 2>&1
  rm -rf $out/t_$name $out/m_$name $out/f_$name $out/w_$name
  cat $out/bench_$name.json; echo; cat $out/kernel_stats_$name.md
done
