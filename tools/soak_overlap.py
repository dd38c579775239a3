"""Soak of the two-queue mode: the C3 stream (F frames) stepped R times on two queues — without any device drain, d2h dma and none alternating,
the host's steps delayed by random 0-100 / 0-300 us in two runs of three — against ONE single-queue run: every frame's triangles and the final map bit for bit.  A race between a frame's
front end and its predecessor's extract would show up as a differing frame sooner or later.
Usage: python tools/soak_overlap.py [R=20] [F=40]"""
import gc, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from di_fusion_amd import synthetic as S
from di_fusion_amd.network import utility as net_util
from di_fusion_amd.stream import FusionStream

DEV = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 20
F = int(sys.argv[2]) if len(sys.argv) > 2 else 40
model = net_util.networks_from_arrays(net_util.load_weights_npz())
scene, cfg = S.config_c3()


import random


def run(overlap, d2h, jitter=0.0, seed=0):
    st = FusionStream(model, scene, cfg, S.Intrinsic(), DEV, F, deg_per_frame=0.5)
    if overlap:
        assert st.enable_overlap()
    outs = []
    rng = random.Random(seed)
    for i in range(F):
        if jitter > 0.0 and i > 2:                       # (moves the front end's start against the previous frame's extract: other interleavings)
            t_end = time.perf_counter() + rng.random() * jitter
            while time.perf_counter() < t_end:
                pass
        o = (st.step_pipelined if i < 2 else st.step_direct)(i, d2h)
        if o is not None:
            if d2h == "none" or i <= 2:
                torch.cuda.synchronize()
            outs.append(tuple(x.clone().cpu() for x in o))
    o = st.flush(d2h)
    torch.cuda.synchronize()
    outs.append(tuple(x.clone().cpu() for x in o))
    n = st.map.n_occupied
    final = (st.map.indexer.clone().cpu(), st.map.latent_vecs[:n].clone().cpu(), st.map.voxel_obs_count[:n].clone().cpu())
    del st
    gc.collect()                                         # (a stream's map and mesh cache refer to each other: their buffers otherwise pile up from run to run)
    return outs, final


ref_outs, ref_final = run(False, "dma")
assert len(ref_outs) == F
bad = 0
t0 = time.time()
for r in range(R):
    d2h = ("dma", "none")[r & 1]
    outs, final = run(True, d2h, jitter=(0.0, 100e-6, 300e-6)[r % 3], seed=r)
    ok = len(outs) == F and all(all(torch.equal(x, y) for x, y in zip(a, b)) for a, b in zip(ref_outs, outs)) and all(torch.equal(x, y) for x, y in zip(ref_final, final))
    if not ok:
        bad += 1
        first = next((f for f, (a, b) in enumerate(zip(ref_outs, outs)) if not all(torch.equal(x, y) for x, y in zip(a, b))), None)
        print(f"repeat {r} (d2h {d2h}): DIFFERS (first differing frame: {first}, frames {len(outs)})", flush=True)
print(f"two-queue soak: {R} runs of {F} C3 frames against the single-queue run: {bad} differ ({time.time() - t0:.0f} s)")
sys.exit(1 if bad else 0)
