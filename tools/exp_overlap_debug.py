"""Where does an overlapped `none`-mode run differ from the eager run?  (ov on/off) x (forced log compaction yes/no) x repeats."""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from di_fusion_amd import synthetic as S
from di_fusion_amd.network import utility as net_util
from di_fusion_amd.stream import FusionStream

DEV = torch.device("cuda:0")
model = net_util.networks_from_arrays(net_util.load_weights_npz())
F = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def make(cap=1 << 13):
    cfg = S.MapConfig((-3.2, -3.2, -3.2), (3.2, 3.2, 3.2), 0.2)
    return FusionStream(model, S.default_room(), cfg, S.Intrinsic().scaled(0.25), DEV, F, deg_per_frame=6.0, initial_capacity=cap)


st = make()
ref = []
for i in range(F):
    o = st.step(i, d2h="new"); torch.cuda.synchronize(); ref.append(tuple(x.clone() for x in o))
ref_lat = st.map.latent_vecs[:st.map.n_occupied].clone()
for ov in (0, 1):
    for gc_at in (0, 3):
        for d2h in ("none", "dma"):
            for rep in range(3):
                st = make()
                if ov:
                    assert st.enable_overlap()
                got = []
                st.step(0, d2h="new"); torch.cuda.synchronize(); got.append(ref[0])
                for i in range(1, F):
                    if gc_at and i == gc_at:
                        st.map._gc_wanted = True
                    o = st.step_direct(i, d2h=d2h)
                    if o is not None:
                        if d2h == "none":
                            torch.cuda.synchronize()
                        got.append(tuple(x.clone() for x in o))
                o = st.flush(d2h); torch.cuda.synchronize(); got.append(tuple(x.clone() for x in o))
                bad = []
                for f, (a, b) in enumerate(zip(ref, got)):
                    if a[0].shape != b[0].shape:
                        bad.append((f, "shape", tuple(a[0].shape), tuple(b[0].shape)))
                    elif not all(torch.equal(x.cpu(), y.cpu()) for x, y in zip(a, b)):
                        bad.append((f, "values", float((a[0].cpu() - b[0].cpu()).abs().max()), int((a[1].cpu() != b[1].cpu()).sum())))
                n = st.map.n_occupied
                lat_ok = n == ref_lat.shape[0] and torch.equal(st.map.latent_vecs[:n], ref_lat)
                print(f"ov={ov} gc_at={gc_at} d2h={d2h} rep={rep}: {'OK' if not bad and lat_ok else 'MISMATCH'} latents {'ok' if lat_ok else 'DIFFER'} {bad[:4]}", flush=True)
