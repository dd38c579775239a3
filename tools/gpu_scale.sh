#!/bin/bash
# The 8-GPU day in one command: the scaling curve of both multi-GPU modes and the one test that needs two GPUs.
#   bash tools/gpu_scale.sh [outdir] [steps]
# For N in 1 2 4 8 (up to the GPUs visible): `bench.py --gpus N` in c4 mode (independent subsequences, weak scaling; once with RCCL brought up behind
# the clock and once — c4rccl — with RCCL alive during the timed region), the same with 4 streams per
# GPU, and in tiled mode (one 1280x960 stream, x-slabs, halo exchange over RCCL, strong scaling); checks that `rccl_ranks == N` for N > 1 and that
# the N = 1 line agrees with the plain 1-GPU bench to +-3 %; prints the curves; then runs tests/test_gpu_parallel.py::test_spatial_tiling_two_processes_rccl.
# Nothing here is measured on a one-GPU box (N stops at the number of visible GPUs).
out=${1:-gpurun_out/scale}; steps=${2:-200}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
export HSA_ENABLE_IPC_MODE_LEGACY=0
ngpu=$(python -c "import torch; print(torch.cuda.device_count())")
echo "visible GPUs: $ngpu"
timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps $steps > $out/ref_n1.json 2> $out/ref_n1.err
port=29600
for mode in c4 c4rccl c4s4 tiled; do
  for N in 1 2 4 8; do
    [ $N -gt $ngpu ] && continue
    extra="--mode c4"; [ $mode = c4rccl ] && extra="--mode c4 --rccl-before-clock 1"; [ $mode = c4s4 ] && extra="--mode c4 --streams-per-gpu 4"; [ $mode = tiled ] && extra="--mode tiled"
    port=$((port + 1))
    if [ $N -eq 1 ]; then
      timeout 900 python bench.py --gpus 1 $extra --no-cpu-baseline --no-secondary --steps $steps > $out/${mode}_n$N.json 2> $out/${mode}_n$N.err
    else
      timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $port bench.py --gpus $N $extra \
        --no-cpu-baseline --no-secondary --steps $steps > $out/${mode}_n$N.json 2> $out/${mode}_n$N.err
    fi
  done
done
python - $out <<'PY'
import json, sys, glob, os
out = sys.argv[1]
def line(f):
    try:
        return json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        return None
ref = line(f"{out}/ref_n1.json")
ok = True
for mode in ("c4", "c4rccl", "c4s4", "tiled"):
    base = None
    print(f"--- {mode} ---")
    for N in (1, 2, 4, 8):
        d = line(f"{out}/{mode}_n{N}.json")
        if d is None:
            continue
        v = d["value"]
        base = base or v
        flags = []
        # rccl_ranks is OBSERVED by bench.py (an all-reduce of a device tensor over the RCCL group returned the right sum), not the launcher's WORLD_SIZE
        if N > 1 and d.get("rccl_ranks") != N:
            flags.append(f"rccl_ranks={d.get('rccl_ranks')} != {N} ({d.get('rccl_probe')})"); ok = False
        merge = (d.get("config") or {}).get("global_map_merge_after_the_clock")
        if N > 1 and mode != "tiled" and (not isinstance(merge, dict) or "error" in merge):
            flags.append(f"global map merge failed: {merge}"); ok = False
        want_before = mode in ("c4rccl", "tiled")
        if N > 1 and (d.get("config") or {}).get("rccl_before_clock") != want_before:
            flags.append(f"rccl_before_clock={(d.get('config') or {}).get('rccl_before_clock')}, wanted {want_before}"); ok = False
        if N == 1 and mode == "c4" and ref and abs(v / ref["value"] - 1) > 0.03:
            flags.append(f"N=1 differs from the plain bench by {100 * (v / ref['value'] - 1):+.1f} %"); ok = False
        eff = v / (base * N) if d.get("scaling") == "weak" else v / base
        print(f"N={N}: {v:10.1f} frames/s  x{v / base:5.2f}  ({'efficiency' if d.get('scaling') == 'weak' else 'speed-up'} {eff:.2f})  {d['ms_per_step']} ms/step  {' ; '.join(flags)}")
print("CHECKS", "ok" if ok else "FAILED")
PY
timeout 900 python -m pytest tests/test_gpu_parallel.py -m gpu -q -k two_processes_rccl 2>&1 | tail -3
