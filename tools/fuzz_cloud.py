"""Differential fuzzing of the point-cloud neighbourhood ops against the exhaustive C oracle: random cloud shapes (uniform, clustered,
lattice with exact ties, duplicates, collinear), sizes, k, radii, strides, NaN rows.  Indices / squared distances / outlier mask bit-exact.
Usage: python tools/fuzz_cloud.py [--cases 30] [--seed 0]"""
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
    for case in range(a.cases):
        n = int(rng.choice([1, 7, 40, 300, 2500, 12000]))
        kind = str(rng.choice(["uniform", "clusters", "lattice", "dups", "line"]))
        scale = float(rng.choice([0.05, 0.5, 3.0]))
        if kind == "uniform":
            p = rng.uniform(-scale, scale, (n, 3))
        elif kind == "clusters":
            c = rng.uniform(-scale, scale, (max(1, n // 50), 3))
            p = c[rng.integers(0, c.shape[0], n)] + rng.normal(scale=scale * 0.01, size=(n, 3))
        elif kind == "lattice":
            m = int(np.ceil(n ** (1 / 3)))
            g = np.stack(np.meshgrid(*[np.arange(m)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n]
            p = g * (scale / 8)
        elif kind == "dups":
            base = rng.uniform(-scale, scale, (max(1, n // 4), 3))
            p = base[rng.integers(0, base.shape[0], n)]
        else:
            p = np.outer(rng.uniform(-scale, scale, n), [1.0, 0.5, -0.25])
        p = p.astype(np.float32)
        if n > 10 and rng.random() < 0.4:
            p[rng.choice(n, max(1, n // 40), replace=False)] = np.nan
        stride = int(rng.choice([3, 4]))
        if stride == 4:
            p = np.concatenate([p, np.zeros((n, 1), np.float32)], 1)
        p = np.ascontiguousarray(p)
        k = int(rng.choice([1, 3, 8, 16, 17, 32]))
        radius = float(scale * rng.choice([0.02, 0.1, 0.5, 4.0]))
        t = torch.from_numpy(p).to(dev)
        idx, dist = ext.knn_search(t, k, radius)
        oi, od = O.knn_bruteforce(p, k, radius)
        assert np.array_equal(idx.cpu().numpy(), oi), (case, kind, n, k, radius, "idx")
        assert np.array_equal(dist.cpu().numpy(), od), (case, kind, n, k, radius, "dist")
        mask = ext.remove_radius_outlier(t, k, radius).cpu().numpy()
        assert np.array_equal(mask, O.remove_radius_outlier(p, k, radius)), (case, kind, n, k, radius, "mask")
        if k >= 6:
            got = ext.estimate_normals(t, k, radius, [0.1, -0.2, 0.3]).cpu().numpy()
            want = O.estimate_normals(p, k, radius, [0.1, -0.2, 0.3])
            assert np.array_equal(np.isnan(got[:, 0]), np.isnan(want[:, 0])), (case, kind, n, k, radius, "normal nan pattern")
        print(f"case {case}: {kind} n={n} stride={stride} k={k} r={radius:.4g} ok", flush=True)
    print("fuzz ok")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=30)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    run(a.cases, a.seed)


if __name__ == "__main__":
    main()
