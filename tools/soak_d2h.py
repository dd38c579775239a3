#!/usr/bin/env python3
"""The two per-frame delivery paths of a directly launched stream against each other on the bench's own workload: C3 (128^3 grid, 640x480),
`frames` frames, once with the export carried by the next frame's kernels (d2h "new") and once with the runtime's copy beside them ("dma").
Every frame's triangles, ids and stds and the final map must be identical.   Usage: python tools/soak_d2h.py [--frames 150]"""
import argparse
import sys
import zlib
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def run(mode, frames):
    from di_fusion_amd import synthetic as syn
    from di_fusion_amd.network import utility as U
    from di_fusion_amd.stream import FusionStream
    dev = torch.device("cuda:0")
    model = U.networks_from_arrays(U.load_weights_npz())
    scene, cfg = syn.config_c3()
    st = FusionStream(model, scene, cfg, syn.Intrinsic(), dev, frames, deg_per_frame=0.5)
    sums = []

    def take(o):
        if o is not None:
            sums.append((int(o[0].shape[0]),) + tuple(zlib.crc32(x.numpy().tobytes()) for x in o))
    o = st.step(0, d2h="new")
    torch.cuda.synchronize()
    take(o)
    for i in range(1, frames):
        take(st.step_direct(i, d2h=mode))
    take(st.flush(mode))
    m = st.map
    n = m.n_occupied
    final = (n, zlib.crc32(m.latent_vecs[:n].cpu().numpy().tobytes()), zlib.crc32(m.indexer.cpu().numpy().tobytes()))
    return sums, final


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=150)
    a = ap.parse_args()
    s_new, f_new = run("new", a.frames)
    s_dma, f_dma = run("dma", a.frames)
    assert len(s_new) == len(s_dma) == a.frames, (len(s_new), len(s_dma))
    bad = [i for i, (x, y) in enumerate(zip(s_new, s_dma)) if x != y]
    assert not bad, f"frames that differ: {bad[:10]}"
    assert f_new == f_dma
    print(f"{a.frames} frames: every frame's triangles / ids / stds and the final map identical under both delivery paths "
          f"({sum(s[0] for s in s_new)} triangles delivered, {f_new[0]} voxels)")
