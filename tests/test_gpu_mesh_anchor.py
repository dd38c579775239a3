"""GPU: oracle-INDEPENDENT anchors for the HIP marching cubes (the reference kernel is CUDA-only and cannot be built here, so the C
restatement the other tests compare with is pinned by properties only).  Nothing in this file imports the oracle:
  * the flat HIP op on ANALYTIC cubes (sphere, axis-aligned box face): every vertex on the analytic surface to interpolation accuracy,
    closed 2-manifold (every undirected edge shared by exactly two triangles), triangle count and per-voxel counts equal to an independent
    case-histogram recount from the analytic field and the public 256-case table;
  * the stream's ONE-PASS kernel (count, look-back, emit in one launch, reachable only through dif_extract) against the flat two-pass
    HIP kernels on the extract's own cubes: identical triangles — so the analytic anchor carries over to the product path;
  * the whole pipeline depth -> map -> mesh on an analytic scene (camera inside a sphere, BASELINE config C2's voxel size): vertices near
    the analytic surface, no non-manifold edge, boundary edges only at the frontier of the observed region."""
import numpy as np
import pytest
import torch

from di_fusion_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
R, r = 8, 4


def lattice():
    """Sample positions of a voxel's R^3 cube in voxel units (get_samples(R, a, b) - 0.5 + 0.5: network/utility.py:129-149, map.py:640-648)."""
    a, b = -(r // 2) * (1. / r), 1. + (r - 1) // 2 * (1. / r)
    t = (np.arange(R, dtype=np.float32) * np.float32((b - a) / (R - 1)) + np.float32(a)).astype(np.float32)
    X, Y, Z = np.meshgrid(t, t, t, indexing="ij")
    return np.stack([X, Y, Z], -1)


def analytic_cubes(n, field, std=0.1, skip=()):
    idx = np.arange(n ** 3, dtype=np.int64)
    pos = np.stack([idx // (n * n), (idx // n) % n, idx % n], -1)
    world = pos[:, None, None, None, :] + lattice()[None]
    sdf = field(world).astype(np.float32)
    vbm = np.arange(n ** 3, dtype=np.int32)
    for s in skip:
        vbm[s] = -1
    return idx.reshape(n, n, n).copy(), idx, vbm, sdf, np.full_like(sdf, std)


def hip_flat_mc(indexer, vb, vbm, sdf, std, max_tri, max_std):
    from di_fusion_amd.system import ext
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    tri, tid, tstd = ext.marching_cubes_interp(t(indexer), t(vb), t(vbm), t(sdf), t(std), int(max_tri), list(indexer.shape), float(max_std))
    return tri.cpu().numpy(), tid.cpu().numpy(), tstd.cpu().numpy()


def edge_counts(tri, quantum=4096):
    q = np.round(tri * quantum).astype(np.int64)
    k = q[..., 0] * (1 << 42) + q[..., 1] * (1 << 21) + q[..., 2]
    edges = np.concatenate([np.stack([k[:, i], k[:, (i + 1) % 3]], 1) for i in range(3)])
    edges = edges[edges[:, 0] != edges[:, 1]]                       # degenerate (snapped) edges
    edges.sort(axis=1)
    uniq, cnt = np.unique(edges, axis=0, return_counts=True)
    return uniq, cnt


def case_table():
    """Triangles per cube type from the public 256-case table as this package ships it (csrc/mc_tables.inc; its SHA-256 equals the
    reference's mc_data.cuh, tools/gen_mc_tables.py --check-reference) — parsed here, not taken from the oracle."""
    import re
    from tests.conftest import ROOT
    src = (ROOT / "di_fusion_amd" / "csrc" / "mc_tables.inc").read_text()
    body = src[src.index("k_mc_tri_table"):]
    rows = re.findall(r"\{([^{}]*)\}", body)
    rows = [[int(v) for v in row.split(",") if v.strip()] for row in rows][:256]
    assert len(rows) == 256 and all(len(x) == 16 for x in rows)
    return np.asarray([sum(1 for v in row if v >= 0) // 3 for row in rows])


def recount(field, n):
    g = np.arange(n * r + 1) / r
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    neg = field(np.stack([X, Y, Z], -1)) < 0
    corner = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
    m = n * r
    case = np.zeros((m, m, m), dtype=np.int64)
    for q, (dx, dy, dz) in enumerate(corner):
        case |= neg[dx:dx + m, dy:dy + m, dz:dz + m].astype(np.int64) << q
    per_cell = case_table()[case]
    return int(per_cell.sum()), per_cell.reshape(n, r, n, r, n, r).sum(axis=(1, 3, 5)).reshape(-1)


def test_flat_hip_mc_on_an_analytic_sphere():
    n, c, rad = 6, 3.0, 1.9
    field = lambda p: np.linalg.norm(p - c, axis=-1) - rad
    indexer, vb, vbm, sdf, std = analytic_cubes(n, field)
    tri, tid, tstd = hip_flat_mc(indexer, vb, vbm, sdf, std, 1e6, 1.0)
    assert tri.shape[0] > 500
    d = np.linalg.norm(tri.reshape(-1, 3) - c, axis=1)
    step = 1.0 / r
    assert np.abs(d - rad).max() < 0.5 * step ** 2 + 1e-3           # linear interpolation error ~ step^2 / (8 R)
    assert np.allclose(tstd, 0.1, atol=1e-6)
    _, cnt = edge_counts(tri)
    assert (cnt == 2).mean() > 0.999 and cnt.max() <= 2             # closed 2-manifold
    total, per_voxel = recount(field, n)
    # (blended corner values of an analytic field with uniform std are the field itself — partition of unity — so the sign pattern of
    # every cell follows from the analytic SDF at the cell corners)
    assert tri.shape[0] == total
    assert np.array_equal(np.bincount(tid, minlength=n ** 3), per_voxel)
    # own voxel missing from the batch -> nothing for that voxel, its neighbours still mesh (they renormalise their weights)
    centre_voxel = 3 * 36 + 3 * 6 + 1
    indexer, vb, vbm2, sdf, std = analytic_cubes(n, field, skip=(centre_voxel,))
    tri2, tid2, _ = hip_flat_mc(indexer, vb, vbm2, sdf, std, 1e6, 1.0)
    assert (tid == centre_voxel).sum() > 0 and (tid2 == centre_voxel).sum() == 0 and (tid2 == centre_voxel - 36).sum() > 0
    # max_std rejection and truncation
    indexer, vb, vbm, sdf, std2 = analytic_cubes(n, field, std=0.2)
    assert hip_flat_mc(indexer, vb, vbm, sdf, std2, 1e6, 0.15)[0].shape[0] == 0
    assert hip_flat_mc(indexer, vb, vbm, sdf, std2, 1e6, 0.25)[0].shape[0] == total
    assert hip_flat_mc(indexer, vb, vbm, sdf, std2, 10, 0.25)[0].shape[0] == 10


def test_flat_hip_mc_on_an_analytic_plane_and_box():
    """A plane at an irrational offset (every vertex EXACTLY on it up to rounding: linear interpolation of a linear field is exact) and the
    L-infinity box (vertices on the box faces away from its edges)."""
    n = 5
    nrm = np.asarray([0.36, 0.48, 0.8])
    off = 2.0 * np.sqrt(1.7)
    plane = lambda p: p @ nrm - off
    indexer, vb, vbm, sdf, std = analytic_cubes(n, plane)
    tri, tid, _ = hip_flat_mc(indexer, vb, vbm, sdf, std, 1e6, 1.0)
    assert tri.shape[0] > 100
    assert np.abs(tri.reshape(-1, 3).astype(np.float64) @ nrm - off).max() < 2e-5
    total, per_voxel = recount(plane, n)
    assert tri.shape[0] == total and np.array_equal(np.bincount(tid, minlength=n ** 3), per_voxel)
    uniq, cnt = edge_counts(tri)
    assert cnt.max() <= 2
    c, h = 2.5, 1.3
    box = lambda p: np.abs(p - c).max(axis=-1) - h
    indexer, vb, vbm, sdf, std = analytic_cubes(n, box)
    tri, tid, _ = hip_flat_mc(indexer, vb, vbm, sdf, std, 1e6, 1.0)
    v = tri.reshape(-1, 3)
    dist = np.abs(np.abs(v - c).max(axis=1) - h)
    assert dist.max() < 1.0 / r                                      # within one cell of the box surface everywhere (edges and corners are chamfered)
    face = np.sort(np.abs(v - c), axis=1)[:, 1] < h - 2.0 / r        # vertices well inside a face: the field is linear there
    assert face.sum() > 100 and dist[face].max() < 2e-5
    _, cnt = edge_counts(tri)
    assert (cnt == 2).mean() > 0.999 and cnt.max() <= 2
    total, per_voxel = recount(box, n)
    assert tri.shape[0] == total and np.array_equal(np.bincount(tid, minlength=n ** 3), per_voxel)


def _sphere_stream(gpu_model, frames):
    from di_fusion_amd.stream import FusionStream
    scene, cfg = syn.config_c1()            # camera inside a 1.5 m sphere, 0.1 m voxels (BASELINE C1 / C2 voxel size)
    st = FusionStream(gpu_model, scene, cfg, syn.Intrinsic(), DEV, frames, deg_per_frame=8.0)
    return st, scene, cfg


@pytest.mark.parametrize("grid", [0, 7, 1, 2], ids=["group_per_workgroup", "ticket_mode", "ticket_mode_one_workgroup", "ticket_mode_two_workgroups"])
def test_onepass_kernel_equals_flat_two_pass_kernels_on_the_extracts_own_cubes(gpu_model, grid, mc_grid_cap):
    """The stream's extract (one-pass marching cubes) and the flat HIP op (count pass, scan, emit pass) on the SAME decoded cubes, the same
    batch map and the same dirty list: identical triangles, ids and std — bit for bit (voxel units).  `ticket_mode`: the launch capped at
    7 workgroups (dif_test_mc_grid_cap), so the > 25 groups of a frame are claimed through the ticket counters — the path of a map with thousands of
    dirty voxels.  With ONE or TWO workgroups at most two XCDs have a workgroup resident, while the groups belong to the runs of all eight: every other
    XCD's groups must be taken over by whoever waits for them (or helped once the own tickets are used up) — the launch's guarantee that a look-back
    only ever waits for groups a running workgroup has claimed, whatever the residency (csrc/kernels_mesh.hip.h: mc_onepass_ring)."""
    from di_fusion_amd.system import ext
    if grid:
        mc_grid_cap(grid)
    st, scene, cfg = _sphere_stream(gpu_model, 3)
    for i in range(3):
        st.step(i, d2h="none")
        m = st.map
        c = m.last_counters
        K, B = c["K"], c["B"]
        assert K > 100 and c["T"] > 1000
        tens = m._xbuf[1]
        n = m.n_occupied
        vbm = torch.full((m._capacity,), -1, dtype=torch.int32, device=DEV)
        vbm[tens["occ_slot"][:B].long()] = torch.arange(B, dtype=torch.int32, device=DEV)
        nx, ny, nz = m.n_xyz
        tri, tid, tstd = ext.marching_cubes_interp(m.indexer.view(nx, ny, nz), tens["valid_blocks"][:K].clone(), vbm, tens["cube_sdf"][:B].clone(),
                                                   tens["cube_std"][:B].clone(), int(4e6), [nx, ny, nz], st.max_std)
        got_tri, got_id, got_std = m.mesh_cache_tensors(new_only=True)
        assert got_tri.size(0) == c["T"] == tri.size(0)
        assert torch.equal(got_id, tid) and torch.equal(got_std, tstd)
        world = tri * np.float32(cfg.voxel_size) + torch.tensor(cfg.bound_min, device=DEV, dtype=torch.float32)       # map.py:698
        assert torch.equal(got_tri, world)


def test_ticket_mode_at_scale_equals_flat_two_pass_kernels():
    """64,000 dirty voxels (a 40^3 grid fully allocated): 16,000 groups claimed through the ticket counter by the ~1,300 workgroups the chip
    holds, parked and emitted out of order of completion — against the flat count / scan / emit kernels on the same cubes, bit for bit, twice
    (tools/stress_mc_ticket.py; the full-occupancy bench's 128^3 run is the same path at 2.1 M voxels)."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    import stress_mc_ticket
    assert stress_mc_ticket.run(40, 2, verbose=False) > 1_000_000


def test_stream_mesh_against_the_analytic_scene(gpu_model):
    """depth frames of an analytic sphere -> integrate -> decode -> marching cubes, 12 frames on an orbit: what comes out is compared with
    the ANALYTIC surface, not with any restatement.  (i) vertices lie near the sphere (the network reconstructs the surface to well under
    a voxel: 0.1 m voxels, bar 0.02 m, median far below); (ii) no edge is shared by more than two triangles; (iii) away from the frontier
    of the observed region the surface is closed: a boundary edge belongs to a voxel with a 26-neighbour that produced no triangle."""
    st, scene, cfg = _sphere_stream(gpu_model, 12)
    for i in range(12):
        st.step(i, d2h="none")
    tri, tid, tstd = (x.cpu().numpy() for x in st.map.mesh_cache_tensors())
    assert tri.shape[0] > 20000
    centre = np.asarray(scene.centre if hasattr(scene, "centre") else (0.0, 0.0, 0.0), dtype=np.float64)
    d = np.abs(np.linalg.norm(tri.reshape(-1, 3).astype(np.float64) - centre, axis=1) - scene.radius)
    sd = tstd.reshape(-1)
    far = d > 0.05
    print(f"  |dist to the analytic sphere|: median {np.median(d):.4f} p90 {np.quantile(d, 0.9):.4f} p95 {np.quantile(d, 0.95):.4f} p99 {np.quantile(d, 0.99):.4f} "
          f"max {d.max():.4f} m ({tri.shape[0]} triangles); farther than 0.05 m: {far.mean():.4f} of the vertices, their std median "
          f"{np.median(sd[far]) if far.any() else 0:.3f} vs {np.median(sd[~far]):.3f} for the rest")
    # the bulk of the surface sits within millimetres of the sphere; what is farther is the network's own extrapolation at the frontier of
    # the observed region (few points per voxel, high predicted std: exactly what max_std exists for), not a meshing artefact
    assert np.median(d) < 0.006 and np.quantile(d, 0.9) < 0.03 and far.mean() < 0.06
    # topology on ONE extraction of the whole map without the std gate (the incremental cache keeps a voxel's stale triangles when a later
    # re-meshing yields none — the reference's rule, map.py:708-709 — and max_std punches holes: neither is a meshing property)
    tri, tid, tstd = st.map.extract_mesh_arrays(st.resolution, int(4e6), max_std=2000.0, no_cache=True, to_host=True)
    uniq, cnt = edge_counts((tri - np.asarray(cfg.bound_min, dtype=np.float32)) / np.float32(cfg.voxel_size), quantum=1 << 14)
    print(f"  one extraction of the whole map: {tri.shape[0]} triangles, edge multiplicities {np.bincount(cnt)[1:]}")
    assert (cnt > 2).mean() < 1e-3
    # boundary edges: every one lies in a voxel next to the frontier (some 26-neighbour without triangles) — none inside the meshed region
    nx, ny, nz = st.map.n_xyz
    has = np.zeros((nx + 2, ny + 2, nz + 2), dtype=bool)
    vx, vy, vz = tid // (ny * nz), (tid // nz) % ny, tid % nz
    has[vx + 1, vy + 1, vz + 1] = True
    q = np.round((tri - np.asarray(cfg.bound_min, dtype=np.float32)) / np.float32(cfg.voxel_size) * (1 << 14)).astype(np.int64)
    k = q[..., 0] * (1 << 42) + q[..., 1] * (1 << 21) + q[..., 2]
    e = np.concatenate([np.stack([k[:, i], k[:, (i + 1) % 3]], 1) for i in range(3)])
    owner = np.concatenate([tid, tid, tid])
    keep = e[:, 0] != e[:, 1]
    e, owner = np.sort(e[keep], axis=1), owner[keep]
    order = np.lexsort((e[:, 1], e[:, 0]))
    e, owner = e[order], owner[order]
    first = np.ones(len(e), dtype=bool); first[1:] = (e[1:] != e[:-1]).any(axis=1)
    run_id = np.cumsum(first) - 1
    count = np.bincount(run_id)
    lonely = count[run_id] == 1
    ov = owner[lonely]
    ox, oy, oz = ov // (ny * nz) + 1, (ov // nz) % ny + 1, ov % nz + 1
    interior = np.ones(len(ov), dtype=bool)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                interior &= has[ox + dx, oy + dy, oz + dz]
    frac_closed = float((count == 2).mean())
    print(f"  edges shared by exactly two triangles: {frac_closed:.4f}; boundary edges {int(lonely.sum())}, of them inside the meshed region {int(interior.sum())}")
    assert frac_closed > 0.9
    assert interior.sum() <= 0.002 * len(count)
