"""Differential fuzzing of integrate + extract against the oracle on random small problems (grid size, voxel size, pruning on/off, point
order, NaNs, out-of-bounds points, repeated frames).  Bit-exact integer state, latents within 2e-5, dirty / batch counts equal; the frame's new
triangles against the oracle's marching cubes on the GPU's own cubes (ids and order exact, vertices within 1e-5) — in about half of the cases
with the one-pass kernel capped at 1-5 workgroups (dif_test_mc_grid_cap: ticket mode, parked groups); then random point queries (mask exact, values within 5e-5).
Usage: python tools/fuzz_integrate.py [--cases 20] [--seed 0]      (GPU; a few seconds per case, the oracle is the slow side)"""
import argparse
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from di_fusion_amd import _lib, synthetic as syn                # noqa: E402
from di_fusion_amd.network import utility as net_util            # noqa: E402
from di_fusion_amd.system.map import DenseIndexedMap             # noqa: E402
from oracle import difusion_oracle as O                          # noqa: E402


def run(cases: int, seed: int = 0):
    a = argparse.Namespace(cases=cases, seed=seed)
    dev = torch.device("cuda:0")
    raw = net_util.load_weights_npz()
    model = net_util.networks_from_arrays(raw)
    onet = O.OracleNetworks(raw)
    rng = np.random.default_rng(a.seed)
    for case in range(a.cases):
        n = int(rng.choice([8, 12, 16, 24]))
        vs = float(rng.choice([0.05, 0.1, 0.2]))
        half = n * vs / 2
        prune = int(rng.choice([0, 4, 16]))
        cfg = syn.MapConfig((-half,) * 3, (half,) * 3, vs, prune_min_vox_obs=prune)
        m = DenseIndexedMap(model, cfg.namespace(), 29, dev, initial_capacity=1024)
        om = O.OracleMap(onet, cfg.bound_min, cfg.bound_max, cfg.voxel_size, prune_min_vox_obs=prune)
        kind = rng.choice(["sphere", "plane", "blob"])
        mc_grid = int(rng.integers(1, 6)) if rng.random() < 0.5 else 0
        if mc_grid:
            _lib.load().dif_test_mc_grid_cap(mc_grid)
        else:
            _lib.load().dif_test_mc_grid_cap(0)
        for frame in range(int(rng.integers(1, 4))):
            N = int(rng.integers(500, 20000))
            if kind == "sphere":
                d = rng.normal(size=(N, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
                p = d * (0.6 * half) + rng.normal(scale=0.01 * half, size=(N, 3)); nr = d
            elif kind == "plane":
                p = np.stack([rng.uniform(-half, half, N), rng.uniform(-half, half, N), rng.normal(scale=0.02 * half, size=N) + 0.1 * half], 1)
                nr = np.tile([0.0, 0.0, 1.0], (N, 1))
            else:
                p = rng.normal(scale=0.4 * half, size=(N, 3)); nr = rng.normal(size=(N, 3)); nr /= np.linalg.norm(nr, axis=1, keepdims=True)
            if rng.random() < 0.5:
                order = np.argsort(p[:, 0] + 10 * p[:, 1])           # spatially coherent order, like image rows
                p, nr = p[order], nr[order]
            p = p.astype(np.float32); nr = nr.astype(np.float32)
            if rng.random() < 0.5:
                bad = rng.choice(N, N // 50, replace=False)
                p[bad[: len(bad) // 2]] = np.nan
                p[bad[len(bad) // 2:]] += 100.0
            mask = m.integrate_keyframe(torch.from_numpy(p).to(dev), torch.from_numpy(nr).to(dev))
            # NaN / out-of-grid points: the product ignores them, the reference (and so the oracle) would index out of range
            with np.errstate(invalid="ignore"):
                gid = np.ceil((p - np.float32(cfg.bound_min[0])) / np.float32(vs)) - 1          # the grid may have one layer more than n (ceil)
                good = np.isfinite(p).all(1) & (gid >= 0).all(1) & (gid < np.asarray(om.n_xyz)[None, :]).all(1)
            omask = om.integrate_keyframe(p[good], nr[good])
            assert (mask is None) == (omask is None)
            if mask is not None:
                mk = mask.cpu().numpy()
                assert np.array_equal(mk[good], omask) and not mk[~good].any(), (case, frame, "mask")
            k = m.n_occupied
            if k != om.n_occupied:
                Path("gpurun_out").mkdir(exist_ok=True)
                gi = m.indexer.cpu().numpy().reshape(-1)
                np.savez_compressed("gpurun_out/fuzz_fail.npz", p=p, nr=nr, good=good, n=n, vs=vs, prune=prune, gpu_indexer=gi,
                                    oracle_indexer=om.indexer.reshape(-1), gpu_pos=m.latent_vecs_pos[:k].cpu().numpy())
                raise AssertionError((case, frame, kind, n, vs, prune, k, om.n_occupied))
            assert np.array_equal(m.indexer.cpu().numpy().reshape(-1), om.indexer.reshape(-1)), (case, frame, "indexer")
            assert np.array_equal(m.voxel_obs_count[:k].cpu().numpy(), om.voxel_obs_count[:k]), (case, frame, "obs")
            if k:
                assert np.abs(m.latent_vecs[:k].cpu().numpy() - om.latent_vecs[:k]).max() < 2e-5, (case, frame, "latent")
            out = m.extract_mesh_arrays(4, int(4e6), max_std=0.15, to_host=False)
            oa = om.extract_prepare(4)
            if oa is None:
                assert m.last_counters["K"] == 0, (case, frame, "K")
            else:
                assert m.last_counters["K"] == len(oa["valid_blocks"]) and m.last_counters["B"] == len(oa["occupied_vec_id"]), (case, frame, "K/B")
                B = m.last_counters["B"]
                tens = m._xbuf[1]
                cs, cd = tens["cube_sdf"][:B].cpu().numpy(), tens["cube_std"][:B].cpu().numpy()
                wt, wi, ws = O.marching_cubes_interp(oa["indexer"], oa["valid_blocks"], oa["vec_batch_mapping"], cs, cd, int(4e6), om.n_xyz, 0.15)
                new = m.mesh_cache_tensors(new_only=True)
                nt = 0 if new is None else new[0].size(0)
                assert nt == wt.shape[0] == m.last_counters["T"], (case, frame, "T", nt, wt.shape[0], mc_grid)
                if nt:
                    assert np.array_equal(new[1].cpu().numpy(), wi), (case, frame, "triangle ids", mc_grid)
                    want = (wt * np.float32(cfg.voxel_size)).astype(np.float32) + om.bound_min
                    assert np.abs(new[0].cpu().numpy() - want).max() < 1e-5 and np.abs(new[2].cpu().numpy() - ws).max() < 1e-5, (case, frame, "vertices", mc_grid)
        # point queries inside the grid (get_sdf, map.py:559-579): validity mask exact, values within 5e-5
        lo = np.asarray(cfg.bound_min, np.float32)
        hi = lo + np.asarray(om.n_xyz, np.float32) * np.float32(vs)
        q = (lo + (hi - lo) * rng.random((4000, 3)).astype(np.float32) * 0.999 + 1e-4).astype(np.float32)
        sdf, std, qm = m.get_sdf(torch.from_numpy(q).to(dev))
        osdf, ostd, oqm = om.get_sdf(q)
        assert np.array_equal(qm.cpu().numpy(), oqm), (case, "query mask")
        if oqm.any():
            assert np.abs(sdf.cpu().numpy() - osdf).max() < 5e-5 and np.abs(std.cpu().numpy() - ostd).max() < 5e-5, (case, "query values")
        print(f"case {case}: grid {n}^3 vs {vs} prune {prune} {kind} mc_grid {mc_grid}: n_occupied {m.n_occupied} triangles {0 if out is None else out[0].shape[0]} ok", flush=True)
    _lib.load().dif_test_mc_grid_cap(0)
    print("fuzz ok")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=20)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    run(a.cases, a.seed)


if __name__ == "__main__":
    main()
