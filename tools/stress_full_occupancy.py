#!/usr/bin/env python3
"""Secondary stress variant of SURVEY.md section 8d: ALL voxels of the C3 grid allocated (128^3 = 2.1 M voxels, 243 MB of latents),
one full extract (decode every voxel's lattice, refine, marching cubes over every voxel).  Reports decoder TFLOP/s at scale
(no tail / launch effects) and marching-cubes throughput.  Latents are those of a real fused frame tiled over the grid, so the
SDF field crosses zero and the refinement + meshing stages carry realistic load.
Usage: python tools/stress_full_occupancy.py [--n 128] [--reps 3]"""
import argparse
import ctypes
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=128)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--max-triangles", type=int, default=int(1.1e8))     # above the ~101 M the 128^3 run produces: every triangle is written
    a = ap.parse_args()
    from di_fusion_amd import _lib, synthetic as syn
    from di_fusion_amd.network import utility as net_util
    from di_fusion_amd.system.map import DenseIndexedMap
    dev = torch.device("cuda:0")
    model = net_util.networks_from_arrays(net_util.load_weights_npz())
    # donor latents: one real frame fused into a small map
    scene, cfg = syn.config_c3()
    donor = DenseIndexedMap(model, cfg.namespace(), 29, dev)
    xyz, nrm = syn.frame_points(scene, 0, syn.Intrinsic(), device=dev)
    donor.integrate_keyframe(xyz, nrm)
    nd = donor.n_occupied
    good = torch.nonzero(donor.voxel_obs_count[:nd] > 16).flatten()
    zl = donor.latent_vecs[:nd][good]
    n = a.n
    half = 0.05 * n / 2
    big_cfg = syn.MapConfig((-half,) * 3, (half,) * 3, 0.05)
    m = DenseIndexedMap(model, big_cfg.namespace(), 29, dev, initial_capacity=n ** 3)
    G = n ** 3
    rec = torch.zeros((G, 32), dtype=torch.int32, device=dev)
    rec[:, 0] = torch.arange(G, device=dev, dtype=torch.int32)
    w = torch.full((G,), 100.0, device=dev)
    rec[:, 2] = w.view(torch.int32)
    pick = torch.randint(0, zl.size(0), (G,), device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    rec[:, 3:32] = (zl[pick] * 100.0).view(torch.int32)
    m.merge_records(rec)
    del rec
    assert m.n_occupied == G
    lib = _lib.load()
    out = []
    for rep in range(a.reps + 1):
        lib.dif_profile_read((ctypes.c_double * _lib.PROF_COUNT)(), (ctypes.c_int64 * _lib.PROF_COUNT)(), 1)
        lib.dif_profile_enable(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.extract_mesh_arrays(4, a.max_triangles, max_std=0.15, no_cache=True, to_host=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        lib.dif_profile_enable(0)
        ms = (ctypes.c_double * _lib.PROF_COUNT)(); nl = (ctypes.c_int64 * _lib.PROF_COUNT)()
        lib.dif_profile_read(ms, nl, 1)
        c = m.last_counters
        prof = {k: ms[i] for i, k in enumerate(_lib.PROF_NAMES)}
        if rep == 0:
            continue
        rows_lat, rows_pts = c["B"] * 64, c["VH"]
        out.append(dict(extract_s=round(dt, 4), K=c["K"], B=c["B"], VH=c["VH"], T=c["T"],
                        decode_lattice_ms=round(prof["decode_lattice"], 3), decode_points_ms=round(prof["decode_points"], 3),
                        decode_lattice_tflops=round(rows_lat * 98816 / (prof["decode_lattice"] * 1e-3) / 1e12, 2),
                        decode_points_tflops=round(rows_pts * 98816 / (prof["decode_points"] * 1e-3) / 1e12, 2),
                        mc_count_ms=round(prof["mc_count"], 3), mc_emit_ms=round(prof["mc_emit"], 3),
                        mc_algorithmic_GBps=round((c["B"] * 2 * 512 * 4 + c["T"] * 56) / ((prof["mc_count"] + prof["mc_emit"]) * 1e-3) / 1e9, 1)))
    print(json.dumps({"workload": f"{n}^3 grid fully allocated ({G} voxels), one extract_mesh(no_cache=True), resolution 4, fast", "runs": out}))


if __name__ == "__main__":
    main()
