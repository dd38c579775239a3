#!/usr/bin/env python3
"""The one-pass marching cubes in ticket mode AT SCALE against the flat two-pass kernels: all voxels of an n^3 grid allocated (latents of a
real fused frame tiled over it, as tools/stress_full_occupancy.py does), one extract — tens of thousands of groups claimed through the ticket
counter by ~1,300 resident workgroups that park and emit them — then the flat HIP op (count pass, scan, emit pass) on the SAME cubes:
identical triangles, ids and stds, bit for bit, in the same order.  Repeats must also equal each other.
Usage: python tools/stress_mc_ticket.py [--n 64] [--reps 3]"""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def run(n=64, reps=3, verbose=True, max_triangles=None):
    max_triangles = int(max_triangles or 60 * n ** 3)           # (~48 triangles per voxel with these latents: nothing is truncated)
    from di_fusion_amd import synthetic as syn
    from di_fusion_amd.network import utility as net_util
    from di_fusion_amd.system import ext
    from di_fusion_amd.system.map import DenseIndexedMap
    dev = torch.device("cuda:0")
    model = net_util.networks_from_arrays(net_util.load_weights_npz())
    scene, cfg = syn.config_c3()
    donor = DenseIndexedMap(model, cfg.namespace(), 29, dev)
    xyz, nrm = syn.frame_points(scene, 0, syn.Intrinsic(), device=dev)
    donor.integrate_keyframe(xyz, nrm)
    nd = donor.n_occupied
    zl = donor.latent_vecs[:nd][torch.nonzero(donor.voxel_obs_count[:nd] > 16).flatten()]
    half = 0.05 * n / 2
    big_cfg = syn.MapConfig((-half,) * 3, (half,) * 3, 0.05)
    m = DenseIndexedMap(model, big_cfg.namespace(), 29, dev, initial_capacity=n ** 3)
    G = n ** 3
    rec = torch.zeros((G, 32), dtype=torch.int32, device=dev)
    rec[:, 0] = torch.arange(G, device=dev, dtype=torch.int32)
    rec[:, 2] = torch.full((G,), 100.0, device=dev).view(torch.int32)
    pick = torch.randint(0, zl.size(0), (G,), device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    rec[:, 3:32] = (zl[pick] * 100.0).view(torch.int32)
    m.merge_records(rec)
    del rec
    assert m.n_occupied == G
    first = None
    for rep in range(reps):
        tri, tid, tstd = m.extract_mesh_arrays(4, max_triangles, max_std=0.15, no_cache=True, to_host=False)
        c = m.last_counters
        K, B, T = c["K"], c["B"], c["T"]
        assert K == G and (K + 3) // 4 > 5 * 256, "ticket mode needs more groups than resident workgroups"
        tens = m._xbuf[1]
        vbm = torch.full((m._capacity,), -1, dtype=torch.int32, device=dev)
        vbm[tens["occ_slot"][:B].long()] = torch.arange(B, dtype=torch.int32, device=dev)
        nx, ny, nz = m.n_xyz
        wt, wi, ws = ext.marching_cubes_interp(m.indexer.view(nx, ny, nz), tens["valid_blocks"][:K].clone(), vbm, tens["cube_sdf"][:B], tens["cube_std"][:B],
                                               max_triangles, [nx, ny, nz], 0.15)
        assert tri.size(0) == T == wt.size(0), (tri.size(0), T, wt.size(0))
        assert torch.equal(tid, wi) and torch.equal(tstd, ws)
        world = wt * np.float32(big_cfg.voxel_size) + torch.tensor(big_cfg.bound_min, device=dev, dtype=torch.float32)       # map.py:698
        assert torch.equal(tri, world)
        if first is None:
            first = (tri.clone(), tid.clone(), tstd.clone())
        else:
            assert all(torch.equal(a, b) for a, b in zip(first, (tri, tid, tstd)))
        if verbose:
            print(f"rep {rep}: {K} voxels, {(K + 3) // 4} groups through the ticket counter, {T} triangles == the flat two-pass kernels, bit for bit", flush=True)
        del wt, wi, ws, world
    return T


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    run(a.n, a.reps)
    print("ticket-mode stress ok")
