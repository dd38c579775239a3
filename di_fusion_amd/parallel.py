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

RECORD_WORDS = 32          # int32 words per voxel record: lin (2) | w (1) | w*z (29)


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
