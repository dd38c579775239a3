"""GPU marching cubes against the C oracle, exhaustively over the kernel's case analysis (the reference kernel,
`ext/marching_cubes/mc_interp_kernel.cu:7-320`, is CUDA-only and cannot be run here, so this leg is pinned by the oracle + this
enumeration + the known-answer properties of tests/test_oracle_mc.py):

  * every one of the 256 cube types, forced into a designated cell of a voxel (consistent corner field), with every neighbour
    present / every neighbour missing / random neighbour sets;
  * every present/missing pattern of the 7 other voxels that feed a corner, for each of the 8 corner octants (8 x 128 patterns),
    plus all-present and all-missing — with consistent fields and with neighbours that DISAGREE (a real std-weighted blend);
  * dirty voxels that are not in the decoded batch (own voxel missing => nothing);
  * `max_std` off (2000) and on (values that reject a good part of the triangles);
  * resolutions r = 1, 2, 3, 4.

Same triangle count, same voxel id per triangle in the same (canonical) order, vertices and per-vertex std within 1e-5 (voxel units)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")

PITCH = 4                   # problems are 3x3x3 voxel blocks at a pitch of 4: one unallocated layer between blocks => no interaction


def neighbour_masks(rng):
    """26-bit presence masks over the 3x3x3 block (bit = (dx+1)*9 + (dy+1)*3 + (dz+1), centre always present)."""
    full = (1 << 27) - 1
    centre = 1 << 13
    out = [full, centre]
    for ox in (-1, 1):
        for oy in (-1, 1):
            for oz in (-1, 1):
                cells = [((dx + 1) * 9 + (dy + 1) * 3 + (dz + 1)) for dx in (0, ox) for dy in (0, oy) for dz in (0, oz)]
                cells.remove(13)
                for pat in range(128):                 # the 7 voxels that share this octant's corners with the centre
                    m = full
                    for b, c in enumerate(cells):
                        if not (pat >> b) & 1:
                            m &= ~(1 << c)
                    out.append(m)
    out += [int(x) | centre for x in rng.integers(0, 1 << 27, 256)]
    return out


def build(r, rng, consistent):
    """One big grid holding all problems of resolution r.  Returns the argument tuple of marching_cubes_interp."""
    R = 2 * r
    masks = neighbour_masks(rng)
    full = (1 << 27) - 1
    probs = [(t, m) for t in range(256) for m in (full, 1 << 13)]
    probs += [(t, int(x) | (1 << 13)) for t in range(256) for x in rng.integers(0, 1 << 27, 4)]
    probs += [(int(rng.integers(0, 256)), m) for m in masks]
    P = len(probs)
    side = int(np.ceil(P ** (1 / 3)))
    n = [PITCH * side] * 3
    G = n[0] * n[1] * n[2]
    # global lattice of multiples of 1/r: every cube sample and every cell corner sits on it
    indexer = -np.ones(G, dtype=np.int64)
    lin_list, cube_s, cube_d = [], [], []
    for p, (ctype, mask) in enumerate(probs):
        px, py, pz = (p // (side * side)) * PITCH, ((p // side) % side) * PITCH, (p % side) * PITCH
        L = 3 * r + 2 * (r // 2) + 2                              # field samples per axis covering the block's cubes
        f_s = rng.choice([-1.0, 1.0], (L, L, L)) * rng.uniform(0.1, 1.0, (L, L, L))
        f_d = rng.uniform(0.05, 0.3, (L, L, L))
        off = r // 2                                              # field index of lattice coordinate 0 of voxel 0 of the block
        # force cube type `ctype` onto one cell of the centre voxel (corner numbering of the kernel / the Bourke table)
        cell = int(rng.integers(0, r ** 3))
        cx, cy, cz = cell // (r * r), (cell // r) % r, cell % r
        for q in range(8):
            dx, dy, dz = int(q in (1, 2, 5, 6)), int(q in (2, 3, 6, 7)), int(q >= 4)
            i, j, k = off + r + cx + dx, off + r + cy + dy, off + r + cz + dz
            f_s[i, j, k] = (-1.0 if (ctype >> q) & 1 else 1.0) * abs(f_s[i, j, k])
        for c in range(27):
            if not (mask >> c) & 1:
                continue
            dx, dy, dz = c // 9, (c // 3) % 3, c % 3
            lin = ((px + dx) * n[1] + (py + dy)) * n[2] + (pz + dz)
            i0, j0, k0 = dx * r, dy * r, dz * r                    # field index of this voxel's first sample (a = -(r//2)/r)
            cs = f_s[i0:i0 + R, j0:j0 + R, k0:k0 + R].copy()
            cd = f_d[i0:i0 + R, j0:j0 + R, k0:k0 + R].copy()
            if not consistent:                                    # neighbours disagree: the blend is a real weighted mean
                cs += rng.normal(scale=0.05, size=cs.shape)
                cd *= rng.uniform(0.7, 1.4, size=cd.shape)
            lin_list.append(lin); cube_s.append(cs); cube_d.append(cd)
    V = len(lin_list)
    lin_arr = np.asarray(lin_list, dtype=np.int64)
    slots = rng.permutation(V)
    indexer[lin_arr] = slots
    # some allocated voxels are not in the decoded batch (vbm = -1), some of those are still listed as dirty: nothing may come out of them
    in_batch = rng.random(V) < 0.97
    vbm = -np.ones(V, dtype=np.int32)
    order = rng.permutation(int(in_batch.sum())).astype(np.int32)
    vbm[slots[in_batch]] = order
    B = int(in_batch.sum())
    cs_all = np.zeros((B, R, R, R), np.float32)
    cd_all = np.zeros((B, R, R, R), np.float32)
    cs_all[order] = np.asarray(cube_s, dtype=np.float32)[in_batch]
    cd_all[order] = np.asarray(cube_d, dtype=np.float32)[in_batch]
    vb = np.sort(np.unique(lin_arr)).astype(np.int64)             # every allocated voxel is dirty, ascending (as the product's lists are)
    return indexer.reshape(n), vb, vbm, cs_all, cd_all, n, P


@pytest.mark.parametrize("r", [1, 2, 3, 4])
@pytest.mark.parametrize("consistent", [True, False])
def test_all_cube_types_and_neighbour_patterns(r, consistent):
    from di_fusion_amd.system import ext
    from oracle import difusion_oracle as O
    rng = np.random.default_rng(100 * r + int(consistent))
    indexer, vb, vbm, cs, cd, n, P = build(r, rng, consistent)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    total, unfiltered = 0, None
    for max_std in (2000.0, 0.2, 0.12):
        wt, wi, ws = O.marching_cubes_interp(indexer, vb, vbm, cs, cd, int(2e7), n, max_std)
        tri, tid, tstd = ext.marching_cubes_interp(t(indexer), t(vb), t(vbm), t(cs), t(cd), int(2e7), n, max_std)
        assert tri.shape[0] == wt.shape[0] > 0, (r, consistent, max_std, tri.shape, wt.shape)
        assert np.array_equal(tid.cpu().numpy(), wi)
        dv = np.abs(tri.cpu().numpy() - wt).max()
        dsd = np.abs(tstd.cpu().numpy() - ws).max()
        assert dv < 1e-5 and dsd < 1e-5, (r, consistent, max_std, dv, dsd)
        total += wt.shape[0]
        if unfiltered is None:
            unfiltered = wt.shape[0]
        else:
            assert wt.shape[0] < unfiltered                      # the std filter really rejected something
    print(f"r={r} consistent={consistent}: {P} problems, {len(vb)} dirty voxels, {total} triangles compared")
