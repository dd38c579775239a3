"""Does a HIGH-priority extracts' stream (the critical chain) beside a normal-priority front-end stream help the two-queue mode?  C3, 200 frames."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from di_fusion_amd import synthetic as S
from di_fusion_amd.network import utility as net_util
from di_fusion_amd.stream import FusionStream

DEV = torch.device("cuda:0")
model = net_util.networks_from_arrays(net_util.load_weights_npz())
scene, cfg = S.config_c3()
F = 220


def run(main_prio, d2h="dma"):
    ctx = torch.cuda.stream(torch.cuda.Stream(device=DEV, priority=main_prio)) if main_prio is not None else None
    if ctx is not None:
        ctx.__enter__()
    try:
        st = FusionStream(model, scene, cfg, S.Intrinsic(), DEV, F, deg_per_frame=0.5)
        assert st.enable_overlap()
        for i in range(20):
            (st.step_pipelined if i < 2 else st.step_direct)(i, d2h)
        st.flush(d2h); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20, F):
            st.step_direct(i, d2h)
        st.flush(d2h); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (F - 20)
    finally:
        if ctx is not None:
            ctx.__exit__(None, None, None)
    print(f"extracts' stream: {'default (null) stream' if main_prio is None else 'pool stream, priority %d' % main_prio}: {dt * 1e6:.1f} us per frame", flush=True)
    del st


for rep in range(2):
    run(None)
    run(0)
    run(-1)
