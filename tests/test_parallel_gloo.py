"""CPU, world_size 2, gloo: the multi-GPU merge path — variable-length all-gather of voxel records, then the same fold on
every rank.  Record arithmetic is played by the oracle here (the HIP kernels need a GPU; tests/test_gpu_parallel.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from di_fusion_amd import synthetic as syn
from tests.conftest import GOLDEN, ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build_rank_map(rank):
    from oracle import difusion_oracle as O
    net = O.OracleNetworks({k: v for k, v in np.load(ROOT / "di_fusion_amd" / "network" / "weights_default.npz").items()})
    cfg = syn.MapConfig((-1.6,) * 3, (1.6,) * 3, 0.4)
    om = O.OracleMap(net, cfg.bound_min, cfg.bound_max, cfg.voxel_size)
    intr = syn.Intrinsic().scaled(0.125)
    for f in range(2):                        # rank r fuses its own arc of the orbit
        xyz, nrm = syn.frame_points(syn.Scene(kind="sphere", radius=1.3), f, intr, deg_per_frame=20.0, phase_deg=90.0 * rank)
        om.integrate_keyframe(xyz.numpy(), nrm.numpy())
    return O, net, cfg, om


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from di_fusion_amd import parallel
    O, net, cfg, om = _build_rank_map(rank)
    rec = torch.from_numpy(O.export_records(om))
    chunks = parallel.all_gather_records(rec)
    assert [c.size(0) for c in chunks][rank] == om.n_occupied
    g = O.OracleMap(net, cfg.bound_min, cfg.bound_max, cfg.voxel_size)
    for c in chunks:
        O.merge_records(g, c.numpy())
    n = g.n_occupied
    q.put((rank, n, g.latent_vecs_pos[:n].copy(), g.voxel_obs_count[:n].copy(), g.latent_vecs[:n].copy(),
           om.n_occupied, om.voxel_obs_count[:om.n_occupied].sum()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_merge_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, n0, pos0, w0, z0, nl0, ws0), (_, n1, pos1, w1, z1, nl1, ws1) = res
    # identical global maps on both ranks, bit for bit, same slot numbering
    assert n0 == n1 and np.array_equal(pos0, pos1) and np.array_equal(w0, w1) and np.array_equal(z0, z1)
    # global = union of voxels, weights add up
    assert n0 >= max(nl0, nl1) and n0 <= nl0 + nl1
    assert abs(w0.sum() - (ws0 + ws1)) < 1e-3
    # first chunk's voxels keep ascending-lin slot order within the chunk
    assert np.all(np.diff(pos0[:nl0]) > 0)


def test_merge_equals_weighted_mean():
    """merge(A, B) at a voxel seen by both = (wA*zA + wB*zB) / (wA + wB)."""
    from oracle import difusion_oracle as O
    _, net, cfg, a = _build_rank_map(0)
    _, _, _, b = _build_rank_map(1)
    g = O.OracleMap(net, cfg.bound_min, cfg.bound_max, cfg.voxel_size)
    O.merge_records(g, O.export_records(a))
    O.merge_records(g, O.export_records(b))
    both = np.intersect1d(a.latent_vecs_pos[:a.n_occupied], b.latent_vecs_pos[:b.n_occupied])
    assert both.size > 0
    for lin in both[:50]:
        sa, sb, sg = a.indexer[lin], b.indexer[lin], g.indexer[lin]
        wa, wb = a.voxel_obs_count[sa], b.voxel_obs_count[sb]
        assert g.voxel_obs_count[sg] == wa + wb
        if wa + wb > 0:
            want = (a.latent_vecs[sa] * wa + b.latent_vecs[sb] * wb) / (wa + wb)
            assert np.abs(g.latent_vecs[sg] - want).max() < 1e-5


class _FakeSlabMap:
    """Stands in for DenseIndexedMap in the exchange plumbing test: records are kept as a dict lin -> (w, z)."""

    def __init__(self, nx, ny, nz, lo, hi):
        self.n_xyz = [nx, ny, nz]
        self._ownership = (lo, hi, 3)
        self.store = {}

    def export_records(self, x_lo, x_hi, raw=False):
        plane = self.n_xyz[1] * self.n_xyz[2]
        rows = []
        for lin in sorted(self.store):
            if x_lo * plane <= lin < x_hi * plane:
                w, z = self.store[lin]
                r = np.zeros(32, dtype=np.int32)
                r[0] = lin; r[2] = np.float32(w).view(np.int32); r[3:32] = z.astype(np.float32).view(np.int32)
                rows.append(r)
        return torch.from_numpy(np.stack(rows) if rows else np.zeros((0, 32), dtype=np.int32))

    def merge_records(self, rec, assign=False):
        assert assign
        for r in rec.numpy():
            self.store[int(r[0])] = (float(r[2:3].view(np.float32)[0]), r[3:32].view(np.float32).copy())


def _halo_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from di_fusion_amd import parallel
    nx = ny = nz = 8
    lo, hi = parallel.slab_range(nx, rank, world)
    m = _FakeSlabMap(nx, ny, nz, lo, hi)
    rng = np.random.default_rng(rank)
    for x in range(lo, hi):                       # every owned voxel of column (y=1,z=2) holds rank-specific data
        m.store[x * ny * nz + 1 * nz + 2] = (100.0 * rank + x, rng.standard_normal(29).astype(np.float32))
    parallel.exchange_halo(m, rank, world)
    q.put((rank, {k: (v[0], v[1].copy()) for k, v in m.store.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_halo_exchange_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_halo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ny = nz = 8
    col = lambda x: x * ny * nz + 1 * nz + 2
    # rank 0 owns x 0..3 and must now also hold rank 1's x = 4,5,6 (its left boundary, 3 layers), bit-exact; and vice versa
    for x in (4, 5, 6):
        assert col(x) in res[0] and res[0][col(x)][0] == res[1][col(x)][0]
        assert np.array_equal(res[0][col(x)][1], res[1][col(x)][1])
    assert col(7) not in res[0]
    for x in (1, 2, 3):
        assert col(x) in res[1] and np.array_equal(res[1][col(x)][1], res[0][col(x)][1])
    assert col(0) not in res[1]
