"""Multi-GPU map merge ("C4": one process per GPU, each fusing its own subsequence into a private
map over the SAME grid; the reference has no distributed code at all).

Fusion is a count-weighted mean, z = sum_i enc(p_i) / sum_i 1, kept as (w, z) per voxel (reference `map.py:448-451`), so
maps merge exactly by adding (w, w*z) records.  The exchange is ONE variable-length all-gather of 128-byte voxel
records over RCCL (`backend="nccl"` is RCCL on ROCm; xGMI underneath) — per keyframe batch / at mesh time, never per frame.
Every rank then folds the gathered chunks, in rank order, into a fresh map: same order everywhere => bit-identical global
maps and identical slot numbering on all ranks.  Tested on CPU with gloo (tests/test_parallel_gloo.py); the record
arithmetic itself runs in HIP (`dif_export_records` / `dif_merge_records`).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

RECORD_WORDS = 32          # int32 words per voxel record: lin | flags | w | payload (29)


def all_gather_records(rec: torch.Tensor, group=None) -> List[torch.Tensor]:
    """Variable-length all-gather of (n_r, 32) int32 record tensors.  Works for nccl (GPU tensors) and gloo (CPU tensors).
    Two collectives: counts (tiny), then the padded payload."""
    assert rec.dtype == torch.int32 and rec.dim() == 2 and rec.size(1) == RECORD_WORDS
    world = dist.get_world_size(group)
    n = torch.tensor([rec.size(0)], dtype=torch.int64, device=rec.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    nmax = max(max(counts), 1)
    padded = torch.zeros((nmax, RECORD_WORDS), dtype=torch.int32, device=rec.device)
    padded[:rec.size(0)] = rec
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded, group=group)
    return [o[:c] for o, c in zip(out, counts)]


def build_global_map(local_map, make_map, group=None):
    """All-gather every rank's voxel records and fold them, in rank order, into the map returned by `make_map()`.
    :return: the global DenseIndexedMap (identical on every rank)."""
    rec = local_map.export_records()
    chunks = all_gather_records(rec, group)
    g = make_map()
    for c in chunks:
        g.merge_records(c)
    return g


# ------------------------------------------------------------------------------------------------------------------
# C5: one stream, the grid cut into x-slabs, one slab per GPU, halo exchange on the slab boundaries
# ------------------------------------------------------------------------------------------------------------------
# x is the slowest-varying axis of the linear voxel id (reference `map.py:292`), so a slab is a contiguous id range.
# Every rank sees the whole frame (640x480 points are 3.7 MB: replicating them is cheaper than routing them) and keeps the
# points whose own voxel lies within HALO voxels of its slab.  Why HALO = 3 makes every OWNED voxel bit-identical to the
# single-map result, provided the boundary layers are refreshed from their owners after every integrate:
#   * an owned voxel v is updated by points whose own voxel u is within 1 of v (the 8-offset gather, map.py:421-429);
#   * such a point is kept only if u holds > prune_min_vox_obs points of the frame (map.py:375)  -> all points of u: dist 1;
#   * and only if u or a 6-neighbour w of u is in the encode set (map.py:392-397): w is within 2 of the slab, and "in the encode
#     set" means allocated (possibly by THIS frame: some new voxel u' within 1 of w, i.e. within 3 of the slab, with all its
#     points and its exact pre-frame indexer entry) and voxel_obs_count[w] < encoder_count_th before this frame's fusion;
#   * sums are order-independent (fixed point), so the same contributions give the same bits.
# Halo voxels are updated locally from incomplete data and are overwritten by the owner's exact (w, z, dirty flag) right after;
# the dirty flag matters because a dirty neighbour pulls ITS neighbourhood into the decoded batch, and marching cubes blends a
# corner over whichever neighbours are in the batch (mc_interp_kernel.cu:17-24).
#
# Exchange volume: a message holds HALO * ny * nz records of 128 bytes (6.3 MB on the 128^3 grid) whatever the occupancy — the price
# of keeping the frame free of host synchronisation (a variable-length transfer needs its length on the host first); at ~153 GB/s
# per xGMI link that is ~40 us per direction, both directions and both neighbours in flight together.
HALO = 3


def slab_range(nx: int, rank: int, world: int):
    return (rank * nx) // world, ((rank + 1) * nx) // world


def exchange_halo(m, rank: int, world: int, group=None, buffers: Optional[dict] = None):
    """Refresh the halo layers of `m` (a map with `set_ownership`) from the neighbouring slabs' owners: one send and one receive per
    neighbour (`ncclSend` / `ncclRecv` grouped by `batch_isend_irecv`; ring neighbours are directly xGMI-linked), nothing else.
    The messages have a fixed size (HALO x-layers, every voxel allocated) and carry their record count in a header row, so the
    exchange needs no host round trip: export, transfers and merge are all enqueued on the stream, and every frame moves the same
    bytes.  `buffers`: dict reused across frames (message tensors), filled on first use."""
    lo, hi = m._ownership[0], m._ownership[1]
    buffers = {} if buffers is None else buffers
    rows = m.halo_message_rows(HALO)
    ops = []
    for name, peer, x0 in (("left", rank - 1, lo), ("right", rank + 1, hi - HALO)):
        if peer < 0 or peer >= world:
            continue
        out = m.export_halo(x0, x0 + HALO, out=buffers.get("out_" + name))       # what that neighbour's halo mirrors
        buffers["out_" + name] = out
        inp = buffers.get("in_" + name)
        if inp is None:
            inp = buffers["in_" + name] = torch.zeros((1 + rows, 32), dtype=torch.int32, device=out.device)
        ops.append(dist.P2POp(dist.isend, out, peer, group))
        ops.append(dist.P2POp(dist.irecv, inp, peer, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()                      # RCCL: orders the current stream behind the transfer; the host does not block
    for name, peer in (("left", rank - 1), ("right", rank + 1)):
        if 0 <= peer < world:
            m.merge_halo(buffers["in_" + name])
    return buffers
