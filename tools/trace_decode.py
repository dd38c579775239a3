"""Phase timestamps inside k_decode_voxels on the C3 stream (instrumented build, -DDIF_TRACE).
Build here (no GPU needed):   python tools/trace_decode.py --build          -> tools/libdifusion_trace.so (git-ignored, ships with gpurun)
Run on the GPU box:           DIF_LIB=tools/libdifusion_trace.so python tools/trace_decode.py --frames 12"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
SO = ROOT / "tools" / "libdifusion_trace.so"


def build():
    from di_fusion_amd import _build
    cmd = [_build.hipcc()] + _build.HIPCC_FLAGS + ["-DDIF_TRACE", str(_build.SOURCES[0]), "-o", str(SO)]
    print(" ".join(cmd))
    subprocess.check_call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--frames", type=int, default=12)
    a = ap.parse_args()
    if a.build:
        build()
        return
    assert os.environ.get("DIF_LIB"), "run with DIF_LIB=tools/libdifusion_trace.so"
    import numpy as np
    import torch
    from di_fusion_amd import _lib, synthetic as syn
    from di_fusion_amd.network import utility as net_util
    from di_fusion_amd.stream import FusionStream
    dev = torch.device("cuda:0")
    scene, cfg = syn.config_c3()
    model = net_util.networks_from_arrays(net_util.load_weights_npz())
    st = FusionStream(model, scene, cfg, syn.Intrinsic(), dev, a.frames, deg_per_frame=0.5)
    lib = ctypes.CDLL(os.environ["DIF_LIB"])
    for i in range(a.frames):
        st.step(i, "none")
    torch.cuda.synchronize()
    B = st.stats[-1]["B"]
    buf = (ctypes.c_ulonglong * (2048 * 8))()
    assert lib.dif_trace_read(buf, 2048 * 8) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 8).astype(np.int64)
    act = t[:, 2] > t[:, 0]                               # waves that decoded a voxel in the LAST launch (stamp 2 sits inside the b < B branch)
    t0 = t[:, 0][t[:, 0] > 0].min()
    us = lambda x: (x - t0) / 100.0
    out = {"B": int(B), "waves_with_work": int(act.sum()),
           "entry_us": [float(np.percentile(us(t[:, 0][t[:, 0] > 0]), q)) for q in (0, 50, 100)],
           "staged_us": [float(np.percentile(us(t[:, 1][t[:, 1] > 0]), q)) for q in (0, 50, 100)],
           "last_voxel_folded_us": [float(np.percentile(us(t[act, 6]), q)) for q in (0, 50, 100)],
           "last_voxel_tiles_done_us": [float(np.percentile(us(t[act, 2]), q)) for q in (0, 50, 100)],
           "last_voxel_upsampled_us": [float(np.percentile(us(t[act, 3]), q)) for q in (0, 50, 100)],
           "last_voxel_listed_us": [float(np.percentile(us(t[act, 4]), q)) for q in (0, 50, 100)],
           "exit_us": [float(np.percentile(us(t[:, 5][t[:, 5] > 0]), q)) for q in (0, 50, 100)],
           "fold_us_median": float(np.median((t[act, 6] - t[act, 1]) / 100.0)),
           "tiles_phase_of_last_voxel_us_median": float(np.median((t[act, 2] - t[act, 6]) / 100.0)),
           "upsample_us_median": float(np.median((t[act, 3] - t[act, 2]) / 100.0)),
           "list_us_median": float(np.median((t[act, 4] - t[act, 3]) / 100.0))}
    print(json.dumps(out))
    # k_encode: 0 entry, 1 staged, 2 last tile's inputs gathered, 3 its MFMA chain done, 4 its records written, 5 exit
    buf = (ctypes.c_ulonglong * (4096 * 8))()          # (the x6 encoder runs 12 waves per workgroup: 3,072 of the 4,096 rows)
    assert lib.dif_trace_read_encode(buf, 4096 * 8) == 0
    e = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
    e0 = e[:, 0][e[:, 0] > 0].min()
    act = (e[:, 2] > e[:, 1]) & (e[:, 1] >= e0)
    ue = lambda x: (x - e0) / 100.0
    print(json.dumps({"kernel": "k_encode", "M": int(st.stats[-1]["M"]), "waves_with_work": int(act.sum()),
                      "staged_us": [float(np.percentile(ue(e[:, 1][e[:, 1] > 0]), q)) for q in (0, 50, 100)],
                      "last_tile_gathered_us": [float(np.percentile(ue(e[act, 2]), q)) for q in (0, 50, 100)],
                      "last_tile_mfma_done_us": [float(np.percentile(ue(e[act, 3]), q)) for q in (0, 50, 100)],
                      "last_tile_written_us": [float(np.percentile(ue(e[act, 4]), q)) for q in (0, 50, 100)],
                      "exit_us": [float(np.percentile(ue(e[:, 5][e[:, 5] > 0]), q)) for q in (0, 50, 100)],
                      "chain_us_median": float(np.median((e[act, 3] - e[act, 2]) / 100.0)),
                      "epilogue_us_median": float(np.median((e[act, 4] - e[act, 3]) / 100.0))}))


if __name__ == "__main__":
    main()
