# usage (on the GPU box, via gpurun): bash tools/gpu_check.sh <tag> [pytest-args...]
# runs the GPU tests, the default bench line and a rocprofv3 kernel-trace of the same command; outputs under gpurun_out/<tag>/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q "$@" > $O/pytest.log 2>&1; tail -15 $O/pytest.log | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json; tail -3 $O/bench.err
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python bench.py --no-cpu-baseline --no-secondary > $O/trace.log 2>&1
python tools/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) --after-nth k_prune_mark 14 --frames 200 > $O/kernel_stats.md 2>&1
find $O/trace -name "*.db" -size +20M -delete
cat $O/kernel_stats.md
