"""CPU: pins the point-cloud neighbourhood oracle.  The reference's ext/pcproc is CUDA only, so the kNN is
pinned against scipy's cKDTree (independent exact implementation) and the closed-form eigenvector against numpy.linalg.eigh."""
import numpy as np
from scipy.spatial import cKDTree

from di_fusion_amd import synthetic as S
from oracle import difusion_oracle as O


def depth_cloud(H=60, W=80, frame=3, noise=0.0, seed=0):
    intr = S.Intrinsic().scaled(W / 640.0)
    assert (intr.height, intr.width) == (H, W)
    R, t = S.orbit_pose(frame, deg_per_frame=5.0)
    depth, _ = S.render_frame(S.default_room(), R, t, intr, noise_seed=seed if noise else None)
    pc = O.unproject_depth(depth.numpy(), intr.fx, intr.fy, intr.cx, intr.cy).reshape(-1, 3)
    pc = pc[~np.isnan(pc[:, 0])]
    return np.concatenate([pc, np.zeros((pc.shape[0], 1), np.float32)], 1).astype(np.float32)      # xyz0 rows (tracker.py:96)


def test_knn_matches_ckdtree():
    pc = depth_cloud()
    k = 16
    idx, dist = O.knn_bruteforce(pc, k)
    d_ref, i_ref = cKDTree(pc[:, :3].astype(np.float64)).query(pc[:, :3].astype(np.float64), k=k)
    np.testing.assert_allclose(dist, d_ref ** 2, rtol=2e-5, atol=1e-9)
    # where consecutive distances are clearly separated the neighbour identity is unambiguous
    gap_ok = np.ones_like(dist, dtype=bool)
    rel = np.abs(np.diff(d_ref ** 2, axis=1)) > 1e-5 * (d_ref[:, 1:] ** 2) + 1e-9
    gap_ok[:, 1:] &= rel
    gap_ok[:, :-1] &= rel
    gap_ok[:, -1] = False                                   # the last entry may tie with the (k+1)-th, which is not in the list
    assert gap_ok.mean() > 0.5
    assert (idx[gap_ok] == i_ref[gap_ok]).all()
    assert (idx[:, 0] == np.arange(pc.shape[0])).all() and (dist[:, 0] == 0).all()
    assert (np.diff(dist, axis=1) >= 0).all()


def test_c_loop_matches_numpy_restatement():
    pc = depth_cloud()[:3000].copy()
    pc[11] = np.nan
    for k, radius in ((16, np.inf), (16, 0.1), (4, 0.03)):
        a, b = O.knn_bruteforce(pc, k, radius), O.knn_bruteforce_numpy(pc, k, radius)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_knn_radius_bound_and_ties():
    pc = depth_cloud()
    idx, dist = O.knn_bruteforce(pc, 16, radius=0.05)
    full_i, full_d = O.knn_bruteforce(pc, 16)
    inside = full_d < np.float32(0.05) ** 2
    assert (idx[inside] == full_i[inside]).all() and (dist[inside] == full_d[inside]).all()
    assert (idx[~inside] == -1).all() and np.isinf(dist[~inside]).all()
    # exact ties resolve to the lower index: a regular lattice has plenty
    g = np.stack(np.meshgrid(np.arange(5), np.arange(5), np.arange(5), indexing="ij"), -1).reshape(-1, 3).astype(np.float32) * 0.5
    i2, d2 = O.knn_bruteforce(g, 7)
    centre = 62
    assert d2[centre, 0] == 0 and (d2[centre, 1:7] == 0.25).all()
    assert (np.diff(i2[centre, 1:7]) > 0).all()


def test_outlier_mask_is_a_radius_count():
    pc = depth_cloud(H=120, W=160)
    rng = np.random.default_rng(1)
    pc[rng.choice(pc.shape[0], 60, replace=False), 2] += rng.uniform(0.1, 0.5, 60).astype(np.float32)     # flying pixels
    mask = O.remove_radius_outlier(pc, 16, 0.05)
    tree = cKDTree(pc[:, :3].astype(np.float64))
    counts = np.array([len(x) for x in tree.query_ball_point(pc[:, :3].astype(np.float64), 0.05)])
    d16 = tree.query(pc[:, :3].astype(np.float64), k=16)[0][:, 15]
    clear = np.abs(d16 - 0.05) > 1e-6
    assert ((counts >= 16) == mask)[clear].all()
    assert 0 < (~mask).sum() < pc.shape[0]


def test_sym3eig_matches_eigh():
    rng = np.random.default_rng(2)
    A = rng.normal(size=(2000, 3, 3)).astype(np.float32)
    scale = np.array([1.0, 0.6, 0.05], np.float32)            # plane-like covariances: smallest eigenvalue well separated
    Q = np.linalg.qr(A.astype(np.float64))[0]
    C = (Q * (scale.astype(np.float64) ** 2)[None, None, :]) @ np.swapaxes(Q, 1, 2)
    C = ((C + np.swapaxes(C, 1, 2)) / 2).astype(np.float32)
    n = O.sym3eig_min(C)
    w, v = np.linalg.eigh(C.astype(np.float64))
    ref = v[:, :, 0]
    np.testing.assert_allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-5)
    assert (np.abs((n * ref).sum(1)) > 1 - 1e-5).all()


def test_normals_on_planes_and_orientation():
    pc = depth_cloud(H=120, W=160)
    nrm = O.estimate_normals(pc, 16, 0.1, [0.0, 0.0, 0.0])
    ok = ~np.isnan(nrm[:, 0])
    assert ok.mean() > 0.9
    np.testing.assert_allclose(np.linalg.norm(nrm[ok], axis=1), 1.0, atol=1e-4)
    assert ((nrm[ok] * pc[ok, :3]).sum(1) <= 1e-6).all()                     # towards the camera at the origin
    # most points lie on a wall / floor / box face of the axis-aligned room seen through a rotated camera: the normal is one
    # of at most three directions up to sign -> |n . n_j| is 0 or 1 for most pairs of points
    sub = nrm[ok][::7]
    dots = np.abs(sub @ sub.T)
    assert ((dots < 0.05) | (dots > 0.95)).mean() > 0.7
    # too few neighbours -> NaN
    sparse = np.concatenate([pc[:50], pc[:3] + 10.0])
    n2 = O.estimate_normals(sparse, 16, 0.1, [0.0, 0.0, 0.0])
    assert np.isnan(n2[-3:]).all()
