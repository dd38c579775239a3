#!/bin/bash
tag=${1:-q}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_map.py tests/test_gpu_pipes.py tests/test_gpu_stream.py tests/test_gpu_parallel.py -m gpu -q -x -s 2>&1 | grep -v "^$\|amdgpu.ids\|Gloo\|socket.cpp" | tail -40
timeout 300 python tools/bench_query.py > $out/bench_query.json 2> $out/bench_query.err; cat $out/bench_query.json; tail -3 $out/bench_query.err
