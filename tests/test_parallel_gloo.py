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


class _OracleMapAdapter:
    """The two methods `parallel.build_global_map` needs, played by the oracle (the product's `DenseIndexedMap` needs a GPU)."""

    def __init__(self, O, om):
        self.O, self.om = O, om

    def export_records(self):
        return torch.from_numpy(self.O.export_records(self.om))

    def merge_records(self, rec):
        self.O.merge_records(self.om, rec.numpy())


def _worker_lists(rank, world, port, q):
    """Two maps per rank (several subsequences per GPU): `build_global_map` with a LIST — one all-gather per list position, fold in
    (rank, position) order."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from di_fusion_amd import parallel
    maps = []
    for j in range(2):
        O, net, cfg, om = _build_rank_map(2 * rank + j)
        maps.append(_OracleMapAdapter(O, om))
    g = parallel.build_global_map(maps, lambda: _OracleMapAdapter(O, O.OracleMap(net, cfg.bound_min, cfg.bound_max, cfg.voxel_size))).om
    n = g.n_occupied
    q.put((rank, n, g.latent_vecs_pos[:n].copy(), g.voxel_obs_count[:n].copy(), g.latent_vecs[:n].copy(),
           [m.om.n_occupied for m in maps], float(sum(m.om.voxel_obs_count[:m.om.n_occupied].sum() for m in maps))))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_merge_two_maps_per_rank_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_lists, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, n0, pos0, w0, z0, nl0, ws0), (_, n1, pos1, w1, z1, nl1, ws1) = res
    assert n0 == n1 and np.array_equal(pos0, pos1) and np.array_equal(w0, w1) and np.array_equal(z0, z1)      # identical on both ranks
    assert abs(w0.sum() - (ws0 + ws1)) < 1e-2                                                                  # every map's weight arrived once
    # the same fold done here, in (rank, position) order = maps 0, 1, 2, 3
    O, net, cfg, _ = _build_rank_map(0)
    g = O.OracleMap(net, cfg.bound_min, cfg.bound_max, cfg.voxel_size)
    for k in range(4):
        O.merge_records(g, O.export_records(_build_rank_map(k)[3]))
    assert g.n_occupied == n0 and np.array_equal(g.latent_vecs_pos[:n0], pos0) and np.array_equal(g.latent_vecs[:n0], z0)


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


class _OracleSlabMap:
    """The surface `parallel.exchange_halo` needs from a map (ownership, fixed-size halo messages in the layout of
    `dif_export_halo` / `dif_merge_halo`), played by the numpy oracle on CPU tensors."""

    def __init__(self, O, om, lo, hi):
        self.O, self.om = O, om
        self.n_xyz = om.n_xyz
        self._ownership = (lo, hi, 3)

    def halo_message_rows(self, layers):
        return layers * self.n_xyz[1] * self.n_xyz[2]

    def export_halo(self, x_lo, x_hi, out=None):
        msg = torch.from_numpy(self.O.export_halo_message(self.om, x_lo, x_hi, self.halo_message_rows(x_hi - x_lo)))
        if out is None:
            return msg
        out.copy_(msg)
        return out

    def merge_halo(self, msg):
        self.O.merge_halo_message(self.om, msg.numpy())

    def integrate(self, xyz, nrm):
        """what `dif_integrate` does under `set_ownership`: points whose own voxel lies outside [lo - halo, hi + halo) are ignored"""
        lo, hi, halo = self._ownership
        ix = np.ceil((xyz[:, 0] - self.om.bound_min[0]) / np.float32(self.om.voxel_size)).astype(np.int64) - 1
        keep = (ix >= lo - halo) & (ix < hi + halo)
        self.om.integrate_keyframe(xyz[keep], nrm[keep])


def _tiling_case():
    from oracle import difusion_oracle as O
    net = O.OracleNetworks({k: v for k, v in np.load(ROOT / "di_fusion_amd" / "network" / "weights_default.npz").items()})
    cfg = syn.MapConfig((-3.2,) * 3, (3.2,) * 3, 0.4)          # 16^3, the case of tests/test_gpu_parallel.py
    frames = [tuple(t.numpy() for t in syn.frame_points(syn.default_room(), f, syn.Intrinsic().scaled(0.25), deg_per_frame=15.0)) for f in range(3)]
    return O, net, cfg, frames


def _halo_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from di_fusion_amd import parallel
    O, net, cfg, frames = _tiling_case()
    om = O.OracleMap(net, cfg.bound_min, cfg.bound_max, cfg.voxel_size)
    m = _OracleSlabMap(O, om, *parallel.slab_range(om.n_xyz[0], rank, world))
    buffers = {}
    for xyz, nrm in frames:
        m.integrate(xyz, nrm)
        parallel.exchange_halo(m, rank, world, buffers=buffers)     # send/recv with the ring neighbours, real message layout
    n = om.n_occupied
    q.put((rank, m._ownership, om.latent_vecs_pos[:n].copy(), om.voxel_obs_count[:n].copy(), om.latent_vecs[:n].copy(), om.updated_vec_id.copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_halo_exchange_gloo():
    """C5 over gloo, world size 2, on the real halo-message layout: two slab maps (oracle arithmetic), the whole frame offered to
    both, neighbour send/recv of the 3 boundary layers after every integrate.  Every OWNED voxel must end up as in the single map:
    same set of voxels, same observation counts bit for bit, latents to fp32 rounding (the oracle's float sums are order-dependent;
    the HIP path's fixed-point sums make this bit-exact, tests/test_gpu_parallel.py), same dirty flags."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_halo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    O, net, cfg, frames = _tiling_case()
    full = O.OracleMap(net, cfg.bound_min, cfg.bound_max, cfg.voxel_size)
    for xyz, nrm in frames:
        full.integrate_keyframe(xyz, nrm)
    nF = full.n_occupied
    posF, wF, zF = full.latent_vecs_pos[:nF], full.voxel_obs_count[:nF], full.latent_vecs[:nF]
    dirtyF = set(posF[full.updated_vec_id].tolist())
    plane = full.n_xyz[1] * full.n_xyz[2]
    covered = 0
    for rank, (lo, hi, _), pos, w, z, upd in res:
        own_full = (posF >= lo * plane) & (posF < hi * plane)
        own_slab = (pos >= lo * plane) & (pos < hi * plane)
        assert set(pos[own_slab].tolist()) == set(posF[own_full].tolist())
        order_f, order_s = np.argsort(posF[own_full]), np.argsort(pos[own_slab])
        assert np.array_equal(w[own_slab][order_s], wF[own_full][order_f])
        assert np.abs(z[own_slab][order_s] - zF[own_full][order_f]).max() < 5e-6
        assert set(p for p in pos[upd].tolist() if lo * plane <= p < hi * plane) == set(p for p in dirtyF if lo * plane <= p < hi * plane)
        # the halo mirrors the neighbour's owned boundary exactly (bit for bit: it is a copy)
        other = res[1 - rank]
        for x in range(max(lo - 3, 0), lo) if rank > 0 else range(hi, min(hi + 3, full.n_xyz[0])):
            mine = (pos >= x * plane) & (pos < (x + 1) * plane)
            theirs = (other[2] >= x * plane) & (other[2] < (x + 1) * plane)
            assert set(pos[mine].tolist()) >= set(other[2][theirs].tolist())
            for lin in other[2][theirs][:40]:
                a, b = np.nonzero(pos == lin)[0][0], np.nonzero(other[2] == lin)[0][0]
                assert w[a] == other[3][b] and np.array_equal(z[a], other[4][b])
        covered += own_full.sum()
    assert covered == nF
