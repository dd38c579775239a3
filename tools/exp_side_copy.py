"""Experiment: does a PCIe copy kernel on a SECOND stream overlap with the frame's kernels for free?  The C3 stream with the mesh left in HBM
(--d2h none), with and without an unrelated 1.43 MB device -> pinned-host copy kernel enqueued on a side stream once per frame."""
import ctypes
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from di_fusion_amd import _lib, synthetic as syn               # noqa: E402
from di_fusion_amd.network import utility as net_util           # noqa: E402
from di_fusion_amd.stream import FusionStream                   # noqa: E402


def run(side, n_tri, warm=5, steps=20):
    dev = torch.device("cuda:0")
    scene, cfg = syn.config_c3()
    model = net_util.networks_from_arrays(net_util.load_weights_npz())
    st = FusionStream(model, scene, cfg, syn.Intrinsic(), dev, warm + steps, deg_per_frame=0.5)
    lib = _lib.load()
    side_stream = torch.cuda.Stream(device=dev, priority=0)
    pins = (torch.empty((n_tri, 3, 3), dtype=torch.float32).pin_memory(), torch.empty((n_tri,), dtype=torch.long).pin_memory(),
            torch.empty((n_tri, 3), dtype=torch.float32).pin_memory())
    for i in range(2):
        st.step(i, "none")
    for i in range(2, warm):
        st.step_direct(i, "none")
    st.flush("none")
    torch.cuda.synchronize()
    b = st.map._cache_struct()
    sp = ctypes.c_void_p(side_stream.cuda_stream)
    t0 = time.perf_counter()
    for i in range(warm, warm + steps):
        st.step_direct(i, "none")
        if side:
            _lib.check(lib.dif_mesh_cache_export(ctypes.byref(b), 0, n_tri, _lib.ptr(pins[0]), _lib.ptr(pins[1]), _lib.ptr(pins[2]), sp), "export")
    st.flush("none")
    torch.cuda.synchronize()
    return steps / (time.perf_counter() - t0)


if __name__ == "__main__":
    out = {}
    for rep in range(2):
        for side in (0, 1):
            out[f"side_copy={side} rep{rep}"] = round(run(bool(side), 25600), 1)
    print(json.dumps(out))
