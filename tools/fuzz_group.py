"""Differential fuzzing of the stream group (`dif_integrate_frames` + `dif_extract_streams`, S independent streams through one chain of launches)
against the same streams stepped alone: random grids (voxel size, prune threshold), frame sizes, stream counts, orbit phases / steps, noise, frame
counts, d2h modes, with a forced mesh-log compaction thrown in.  Everything must be BIT-identical per stream: every frame's new triangles, the final
map, the final mesh cache.  Usage: python tools/fuzz_group.py [--cases 10] [--seed 0]   (GPU)"""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from di_fusion_amd import synthetic as syn                      # noqa: E402
from di_fusion_amd.network import utility as net_util            # noqa: E402
from di_fusion_amd.stream import FusionStream, FusionStreamGroup  # noqa: E402


def snapshot(st):
    m = st.map
    n = m.n_occupied
    t = m.mesh_cache_tensors(new_only=False)
    if t is None:                                             # (nothing has been meshed yet: a coarse grid seen through a tiny frame)
        t = (torch.zeros((0, 3, 3), device=m.device), torch.zeros((0,), dtype=torch.long, device=m.device), torch.zeros((0, 3), device=m.device))
    tri, tid, tstd = t
    return dict(n=n, indexer=m.indexer.clone(), latent=m.latent_vecs[:n].clone(), obs=m.voxel_obs_count[:n].clone(), tri=tri.clone(), tid=tid.clone(),
                tstd=tstd.clone())


def run(cases: int, seed: int = 0):
    dev = torch.device("cuda:0")
    model = net_util.networks_from_arrays(net_util.load_weights_npz())
    rng = np.random.default_rng(seed)
    for case in range(cases):
        vs = float(rng.choice([0.1, 0.2, 0.4]))
        prune = int(rng.choice([4, 16]))
        cfg = syn.MapConfig((-3.2,) * 3, (3.2,) * 3, vs, prune_min_vox_obs=prune)
        intr = syn.Intrinsic().scaled(float(rng.choice([0.125, 0.25])))
        S = int(rng.integers(2, 6))
        F = int(rng.integers(3, 6))
        deg = float(rng.choice([2.0, 6.0, 15.0]))
        noise = bool(rng.integers(0, 2))
        d2h = str(rng.choice(["new", "new", "none"]))
        scene = syn.default_room() if rng.integers(0, 2) else syn.Scene(kind="sphere", radius=1.5)
        phases = [float(rng.uniform(0, 360)) for _ in range(S)]
        gc_at = int(rng.integers(1, F)) if rng.integers(0, 2) else -1
        gc_stream = int(rng.integers(0, S))

        def mk(j):
            return FusionStream(model, scene, cfg, intr, dev, F, deg_per_frame=deg, phase_deg=phases[j], noise=noise)

        def eager(st, i):
            o = st.step(i, d2h=d2h)
            torch.cuda.synchronize()
            return None if (o is None or d2h == "none") else tuple(x.clone() for x in o)

        solo = []
        for j in range(S):
            st = mk(j)
            solo.append(([eager(st, i) for i in range(F)], snapshot(st)))
            del st
        streams = [mk(j) for j in range(S)]
        got = [[eager(st, 0)] for st in streams]
        grp = FusionStreamGroup(streams)
        for i in range(1, F):
            if i == gc_at:
                streams[gc_stream].map._gc_wanted = True
            outs = grp.step(i, d2h=d2h)
            torch.cuda.synchronize()
            for j, o in enumerate(outs):
                if o is not None:
                    got[j].append(None if d2h == "none" else tuple(x.clone() for x in o))
        for j, o in enumerate(grp.flush(d2h)):
            got[j].append(None if (o is None or d2h == "none") else tuple(x.clone() for x in o))
        for j in range(S):
            a, b = solo[j][1], snapshot(streams[j])
            assert a["n"] == b["n"], (case, j)
            for k in ("indexer", "latent", "obs", "tri", "tid", "tstd"):
                assert torch.equal(a[k], b[k]), (case, j, k)
            if d2h == "new":
                assert len(got[j]) == F, (case, j, len(got[j]))
                for f in range(F):
                    x, y = solo[j][0][f], got[j][f]
                    if x is None or y is None:            # (the eager step hands back None while nothing has ever been meshed; the group an empty update)
                        assert (x is None or x[0].shape[0] == 0) and (y is None or y[0].shape[0] == 0), (case, j, f)
                        continue
                    assert all(torch.equal(p, q) for p, q in zip(x, y)), (case, j, f)
        print(f"case {case}: S={S} F={F} vs={vs} prune={prune} {intr.width}x{intr.height} d2h={d2h} noise={noise} gc@{gc_at} "
              f"voxels={[s[1]['n'] for s in solo]} triangles={[int(s[1]['tri'].shape[0]) for s in solo]}  ok", flush=True)
        del streams, grp, solo
        torch.cuda.empty_cache()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=10)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    run(a.cases, a.seed)
