"""Where does the host's time per frame go?  C3 stream, 120 frames per mode: wall time per frame, time inside the two enqueue calls, inside the
completion of the previous frame (stamp wait + export), for d2h none / dma by hipMemcpyAsync / dma by SDMA, one and two queues."""
import sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from di_fusion_amd import _lib, synthetic as S
from di_fusion_amd.network import utility as net_util
from di_fusion_amd import stream as stream_mod
from di_fusion_amd.stream import FusionStream

DEV = torch.device("cuda:0")
model = net_util.networks_from_arrays(net_util.load_weights_npz())
scene, cfg = S.config_c3()
F = 140
lib = _lib.load()


def run(d2h, sdma, ov, depth=1, split=True):
    st = FusionStream(model, scene, cfg, S.Intrinsic(), DEV, F, deg_per_frame=0.5)
    st.host_depth = depth
    st.split_mesh = split
    if ov:
        assert st.enable_overlap()
    if not sdma:
        st._sdma = False
    t_fin, t_exp = [], []
    orig_fin = st._finish_pending
    def fin(d):
        t0 = time.perf_counter(); r = orig_fin(d); t_fin.append(time.perf_counter() - t0); return r
    st._finish_pending = fin
    orig_wait = _lib.spin_until
    t_spin = []
    def spin(*a, **k):
        t0 = time.perf_counter(); orig_wait(*a, **k); t_spin.append(time.perf_counter() - t0)
    stream_mod._lib.spin_until = spin
    import di_fusion_amd.system.map as mp
    mp._lib.spin_until = spin
    for i in range(20):
        (st.step_pipelined if i < 2 else st.step_direct)(i, d2h)
    st.flush(d2h); torch.cuda.synchronize()
    t_fin.clear(); t_spin.clear()
    t0 = time.perf_counter()
    for i in range(20, F):
        st.step_direct(i, d2h)
    st.flush(d2h); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (F - 20)
    stream_mod._lib.spin_until = orig_wait; mp._lib.spin_until = orig_wait
    print(f"d2h={d2h:5s} sdma={int(sdma)} two_queues={int(ov)} depth={depth} split={int(split)}: {dt * 1e6:7.1f} us/frame; completing the previous frame {np.mean(t_fin) * 1e6:6.1f} us "
          f"(of which waiting for its stamp {np.sum(t_spin) / len(t_fin) * 1e6:6.1f}); enqueue + the rest {dt * 1e6 - np.mean(t_fin) * 1e6:6.1f} us", flush=True)


run("dma", True, 0)
for depth in (1, 2):
    for split in (False, True):
        run("dma", True, 1, depth, split)
        run("none", True, 1, depth, split)
