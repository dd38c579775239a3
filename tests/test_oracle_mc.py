"""CPU: known-answer properties pinning the C marching-cubes oracle (the reference kernel is CUDA-only: parity unpinned)."""
import numpy as np

from oracle import difusion_oracle as O

R = 8
r = 4


def lattice():
    a, b = -(r // 2) * (1. / r), 1. + (r - 1) // 2 * (1. / r)
    return O.get_samples(R, a, b).reshape(R, R, R, 3)          # voxel-local coordinates in [-0.5, 1.25]


def sphere_setup(n=6, radius=1.9, centre=(3.0, 3.0, 3.0), std=0.1, skip=()):
    """n^3 grid, every voxel allocated & batched; cube values = analytic sphere SDF (already 'negated' convention: <0 inside)."""
    n_xyz = [n, n, n]
    lat = lattice()
    idx = np.arange(n ** 3, dtype=np.int64)
    pos = np.stack([idx // (n * n), (idx // n) % n, idx % n], -1)
    world = pos[:, None, None, None, :] + lat[None]
    sdf = (np.linalg.norm(world - np.asarray(centre), axis=-1) - radius).astype(np.float32)
    indexer = idx.reshape(n_xyz).copy()
    vbm = np.arange(n ** 3, dtype=np.int32)
    for s in skip:
        vbm[s] = -1
    return n_xyz, indexer, idx, vbm, sdf, np.full_like(sdf, std)


def test_vertices_lie_on_the_sphere_and_surface_is_closed():
    n_xyz, indexer, vb, vbm, sdf, std = sphere_setup()
    tri, tid, tstd = O.marching_cubes_interp(indexer, vb, vbm, sdf, std, int(1e6), n_xyz, 1.0)
    assert tri.shape[0] > 500
    d = np.linalg.norm(tri.reshape(-1, 3) - 3.0, axis=1)
    assert np.abs(d - 1.9).max() < 0.5 * 0.25 ** 2 + 1e-3           # linear interpolation error ~ step^2 / (8 R)
    assert np.allclose(tstd, 0.1, atol=1e-6)
    # closed 2-manifold: every undirected edge is shared by exactly two triangles
    q = np.round(tri * 4096).astype(np.int64)
    key = lambda p: p[..., 0] * (1 << 40) + p[..., 1] * (1 << 20) + p[..., 2]
    k = key(q)
    edges = np.concatenate([np.stack([k[:, i], k[:, (i + 1) % 3]], 1) for i in range(3)])
    edges = edges[edges[:, 0] != edges[:, 1]]                       # degenerate (snapped) edges
    edges.sort(axis=1)
    _, cnt = np.unique(edges, axis=0, return_counts=True)
    assert (cnt == 2).mean() > 0.999


def test_triangle_count_equals_case_histogram():
    """Independent recount: corner signs of every cell -> case id -> table length."""
    n_xyz, indexer, vb, vbm, sdf, std = sphere_setup()
    tri, tid, _ = O.marching_cubes_interp(indexer, vb, vbm, sdf, std, int(1e6), n_xyz, 1.0)
    _, table = O.mc_tables()
    ntri = (table >= 0).sum(1) // 3
    n = n_xyz[0]
    # blended corner values of an analytic field with uniform std are the field itself (partition of unity), so the
    # sign pattern can be recomputed from the analytic SDF at the cell corners
    g = np.arange(n * r + 1) / r
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    f = np.sqrt((X - 3) ** 2 + (Y - 3) ** 2 + (Z - 3) ** 2) - 1.9
    neg = f < 0
    corner = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
    m = n * r
    case = np.zeros((m, m, m), dtype=np.int64)
    for q, (dx, dy, dz) in enumerate(corner):
        case |= neg[dx:dx + m, dy:dy + m, dz:dz + m].astype(np.int64) << q
    assert tri.shape[0] == int(ntri[case].sum())
    # per-voxel counts too
    per_voxel = ntri[case].reshape(n, r, n, r, n, r).sum(axis=(1, 3, 5)).reshape(-1)
    assert np.array_equal(np.bincount(tid, minlength=n ** 3), per_voxel)


def test_missing_own_voxel_emits_nothing_and_neighbours_still_blend():
    centre_voxel = 3 * 36 + 3 * 6 + 1          # a voxel crossed by the sphere
    n_xyz, indexer, vb, vbm, sdf, std = sphere_setup(skip=(centre_voxel,))
    tri, tid, _ = O.marching_cubes_interp(indexer, vb, vbm, sdf, std, int(1e6), n_xyz, 1.0)
    assert (tid == centre_voxel).sum() == 0                          # own cube missing -> NaN corners -> no cells
    full = O.marching_cubes_interp(indexer, vb, np.arange(216, dtype=np.int32), sdf, std, int(1e6), n_xyz, 1.0)
    assert (full[1] == centre_voxel).sum() > 0
    nb = centre_voxel - 36                    # (2,3,1): also crossed by the sphere
    assert (tid == nb).sum() > 0                                     # neighbours renormalise their weights


def test_max_std_rejection_and_truncation():
    n_xyz, indexer, vb, vbm, sdf, std = sphere_setup(std=0.2)
    assert O.marching_cubes_interp(indexer, vb, vbm, sdf, std, int(1e6), n_xyz, 0.15)[0].shape[0] == 0
    full = O.marching_cubes_interp(indexer, vb, vbm, sdf, std, int(1e6), n_xyz, 0.25)[0].shape[0]
    assert full > 0
    assert O.marching_cubes_interp(indexer, vb, vbm, sdf, std, 10, n_xyz, 0.25)[0].shape[0] == 10


def test_std_weighted_blending_prefers_large_std():
    """STD_W_SDF (mc_interp_kernel.cu:32,109): sdf = sum s*w*sigma / sum w*sigma.  Two voxels along x with constant fields
    -1 (sigma 0.1) and +1 (sigma 0.3): on the shared face w = 1/2 each, blended = (-0.1 + 0.3)/(0.4) = 0.5 > 0."""
    n_xyz = [2, 1, 1]
    indexer = np.array([0, 1], dtype=np.int64).reshape(n_xyz)
    sdf = np.stack([np.full((R, R, R), -1.0), np.full((R, R, R), 1.0)]).astype(np.float32)
    std = np.stack([np.full((R, R, R), 0.1), np.full((R, R, R), 0.3)]).astype(np.float32)
    tri, tid, tstd = O.marching_cubes_interp(indexer, np.array([0, 1], dtype=np.int64), np.array([0, 1], dtype=np.int32), sdf, std, 10000, n_xyz, 10.0)
    # zero crossing of the blended field: (-(1-t)*0.1 + t*0.3) = 0 at weight t = 0.25 of voxel 1, i.e. x = 0.5 + 0.25 = 0.75
    xs = tri[..., 0].reshape(-1)
    assert tri.shape[0] > 0 and np.abs(xs - 0.75).max() < 1e-5


def test_empty_inputs():
    n_xyz, indexer, vb, vbm, sdf, std = sphere_setup()
    tri, tid, tstd = O.marching_cubes_interp(indexer, vb[:0], vbm, sdf, std, 100, n_xyz, 1.0)
    assert tri.shape == (0, 3, 3) and tid.shape == (0,) and tstd.shape == (0, 3)
