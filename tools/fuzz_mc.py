"""Differential fuzzing of the flat marching-cubes op against the C oracle: random grids, random allocation / batch membership, smooth
and rough random cubes, several resolutions, max_std values.  Same triangle count, same voxel ids in order, vertices within 1e-5.
Usage: python tools/fuzz_mc.py [--cases 40] [--seed 0]"""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from di_fusion_amd.system import ext                             # noqa: E402
from oracle import difusion_oracle as O                          # noqa: E402


def run(cases: int, seed: int = 0):
    a = argparse.Namespace(cases=cases, seed=seed)
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(a.seed)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    for case in range(a.cases):
        n = [int(rng.integers(2, 7)) for _ in range(3)]
        r = int(rng.choice([1, 2, 3, 4]))
        R = 2 * r
        G = int(np.prod(n))
        alloc = rng.random(G) < rng.choice([0.3, 0.7, 1.0])
        V = int(alloc.sum())
        if V == 0:
            continue
        indexer = -np.ones(G, dtype=np.int64)
        slots = rng.permutation(V)
        indexer[np.nonzero(alloc)[0]] = slots
        in_batch = rng.random(V) < rng.choice([0.5, 0.9, 1.0])
        B = int(in_batch.sum())
        if B == 0:
            continue
        vbm = -np.ones(V, dtype=np.int32)
        vbm[np.nonzero(in_batch)[0]] = rng.permutation(B).astype(np.int32)
        lin_of_slot = np.empty(V, dtype=np.int64)
        lin_of_slot[indexer[alloc]] = np.nonzero(alloc)[0]
        cand = lin_of_slot[np.nonzero(in_batch)[0]]
        K = int(rng.integers(1, len(cand) + 1))
        vb = np.sort(rng.choice(cand, K, replace=False)).astype(np.int64)
        if rng.random() < 0.5:      # smooth field: a random sphere sampled on every voxel's lattice
            c = rng.uniform(0, max(n), 3); rad = rng.uniform(0.8, max(n) / 1.5)
            aa, bb = -(r // 2) * (1. / r), 1. + (r - 1) // 2 * (1. / r)
            lat = O.get_samples(R, aa, bb).reshape(R, R, R, 3)
            pos = np.stack([lin_of_slot // (n[1] * n[2]), (lin_of_slot // n[2]) % n[1], lin_of_slot % n[2]], -1)
            cube_all = (np.linalg.norm(pos[:, None, None, None, :] + lat[None] - c, axis=-1) - rad).astype(np.float32)
            cs = np.zeros((B, R, R, R), np.float32)
            cs[vbm[in_batch]] = cube_all[np.nonzero(in_batch)[0]]
        else:
            cs = rng.normal(scale=0.3, size=(B, R, R, R)).astype(np.float32)
        cd = rng.uniform(0.05, 0.3, size=(B, R, R, R)).astype(np.float32)
        max_std = float(rng.choice([0.12, 0.2, 2000.0]))
        wt, wi, ws = O.marching_cubes_interp(indexer.reshape(n), vb, vbm, cs, cd, int(4e6), n, max_std)
        tri, tid, tstd = ext.marching_cubes_interp(t(indexer.reshape(n)), t(vb), t(vbm), t(cs), t(cd), int(4e6), n, max_std)
        assert tri.shape[0] == wt.shape[0], (case, n, r, tri.shape, wt.shape)
        if wt.shape[0]:
            assert np.array_equal(tid.cpu().numpy(), wi), (case, "ids")
            assert np.abs(tri.cpu().numpy() - wt).max() < 1e-5, (case, "vertices", np.abs(tri.cpu().numpy() - wt).max())
            assert np.abs(tstd.cpu().numpy() - ws).max() < 1e-5, (case, "std")
        print(f"case {case}: grid {n} r={r} V={V} B={B} K={K} max_std={max_std}: {wt.shape[0]} triangles ok", flush=True)
    print("fuzz ok")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    run(a.cases, a.seed)


if __name__ == "__main__":
    main()
