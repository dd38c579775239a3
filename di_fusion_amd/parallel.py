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


def _host_staged(t: torch.Tensor, group=None) -> bool:
    """True when `t` lives on the GPU but the process group cannot move device memory (gloo): the transfer is then staged through pinned
    host memory.  This is the transport of the two-processes-on-ONE-GPU tests (RCCL refuses two ranks on one device) and of hosts
    without xGMI peer access; with `backend="nccl"` (RCCL) device tensors go out as they are."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _pinned_like(t: torch.Tensor, cache: Optional[dict], key) -> torch.Tensor:
    if cache is None:
        return torch.empty(t.shape, dtype=t.dtype).pin_memory()
    h = cache.get(key)
    if h is None or h.shape != t.shape:
        h = cache[key] = torch.empty(t.shape, dtype=t.dtype).pin_memory()
    return h


def all_gather_records(rec: torch.Tensor, group=None) -> List[torch.Tensor]:
    """Variable-length all-gather of (n_r, 32) int32 record tensors.  Works for nccl (GPU tensors), gloo (CPU tensors) and gloo with GPU
    tensors (staged through host memory).  Two collectives: counts (tiny), then the padded payload."""
    assert rec.dtype == torch.int32 and rec.dim() == 2 and rec.size(1) == RECORD_WORDS
    world = dist.get_world_size(group)
    dev = rec.device
    staged = _host_staged(rec, group)
    if staged:
        rec = rec.cpu()
    n = torch.tensor([rec.size(0)], dtype=torch.int64, device=rec.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    nmax = max(max(counts), 1)
    padded = torch.zeros((nmax, RECORD_WORDS), dtype=torch.int32, device=rec.device)
    padded[:rec.size(0)] = rec
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded, group=group)
    return [(o[:c].to(dev) if staged else o[:c]) for o, c in zip(out, counts)]


def build_global_map(local_map, make_map, group=None):
    """All-gather every rank's voxel records and fold them, in rank order, into the map returned by `make_map()`.
    `local_map`: one map, or the list of this rank's maps (several subsequences per GPU, the same number on every rank): one all-gather per
    list position, folded in (rank, position) order.
    :return: the global DenseIndexedMap (identical on every rank)."""
    maps = list(local_map) if isinstance(local_map, (list, tuple)) else [local_map]
    gathered = [all_gather_records(m.export_records(), group) for m in maps]
    g = make_map()
    for r in range(len(gathered[0])):
        for chunks in gathered:
            g.merge_records(chunks[r])
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
# Halo voxels are never written by the rank that merely mirrors them (integrate's gather drops pairs that target a voxel outside the own
# slab): they change only through the owner's messages.  The dirty flag travels too, because a dirty neighbour pulls ITS neighbourhood
# into the decoded batch, and marching cubes blends a corner over whichever neighbours are in the batch (mc_interp_kernel.cu:17-24).
#
# Two kinds of message, both of a size the HOST knows when it posts the transfer (a length that only the device knows would cost a host
# round trip per frame), both carrying their record count in a header row that the merge kernels read on the device:
#   * "full":  all allocated voxels of the HALO boundary layers — HALO * ny * nz records of 128 bytes (6.3 MB on the 128^3 grid)
#              whatever changed; exact from any state of the receiver;
#   * "delta": only the boundary voxels the owner allocated or fused since its last export (`dif_map_t.halo_list`, appended to by the
#              allocation scan and by k_fuse), at most DELTA_ROWS records (512 KB); exact provided the receiver's copy was exact before
#              the frame — which whole-layer messages on the first two frames establish and every exact exchange preserves.
# Which kind a message is must be known to BOTH ends when they post the transfer, without talking: it is a function of the header of the
# message that travelled the same way two frames earlier (both ends have it in pinned host memory by then, written by the export /
# merge kernels themselves): "delta" while the pending changes stayed below DELTA_ROWS / 4, "full" otherwise and on frames 0-1.
# A delta that still overflows (a more than fourfold jump within two frames) raises on both ends — the device flags it (DIF_C_OVERFLOW
# = 8) and the header says so —; such streams run with mode="full".
HALO = 3
DELTA_ROWS = 4096


def slab_range(nx: int, rank: int, world: int):
    return (rank * nx) // world, ((rank + 1) * nx) // world


def restart_halo_exchange(buffers: dict):
    """Forget the exchange history: the next two exchanges use whole-layer messages.  To be called ON EVERY RANK after a tiled map was
    changed by anything but integrate (merge_records, load, latent optimisation): those changes are not in the boundary change lists."""
    # the rotating pinned note slots are zeroed by the host when a frame takes one: every kernel that may still write a header into
    # them (exports / merges of the frames before the restart) has to be done first, or a stale header would be read as frame f's and
    # flip the delta / whole-layer decision on this rank only
    for h in buffers.get("hist", {}).values():
        if not h.get("done") and h.get("event") is not None:
            h["event"].synchronize()
    buffers["frame"] = 0
    buffers["hist"] = {}


def _message_kind(st: dict, f: int, which: str, side: int, rows_full: int, mode: str) -> str:
    if mode == "full" or rows_full <= DELTA_ROWS or f < 2:
        return "full"
    h = st["hist"][f - 2]
    if not h.get("done"):
        h["event"].synchronize()                             # two frames back: long done
        h["done"] = True
    n, pending, was_delta = (int(v) for v in h[which][4 * side:4 * side + 3])
    if was_delta and pending > n:
        raise RuntimeError(f"halo exchange: the delta message of frame {f - 2} overflowed ({pending} changed boundary voxels, room for {n}): "
                           "the halo copies are stale from that frame on; run this stream with whole-layer messages (mode='full')")
    return "delta" if pending <= DELTA_ROWS // 4 else "full"


class HaloExchange:
    """One rank's side of the per-frame halo refresh, in three steps so that an in-process emulation of several slabs (tests, the
    loopback bench) can run them phase by phase: `export()` -> the transfer -> `merge()`.  `exchange_halo` below does all three."""

    def __init__(self, m, rank: int, world: int, state: Optional[dict] = None, mode: str = "delta"):
        self.m, self.rank, self.world, self.mode = m, rank, world, mode
        self.st = {} if state is None else state
        lo, hi = m._ownership[0], m._ownership[1]
        self.sides = [(name, peer, k, x0) for k, (name, peer, x0) in enumerate((("left", rank - 1, lo), ("right", rank + 1, hi - HALO))) if 0 <= peer < world]
        self.rows_full = m.halo_message_rows(HALO)
        dev = getattr(m, "device", None)
        self.on_gpu = dev is not None and torch.device(dev).type == "cuda"
        self.dev = dev if self.on_gpu else "cpu"
        self.rows = {"full": self.rows_full, "delta": DELTA_ROWS}
        self.reserved = False               # the caller has made room in the map for the voxels the incoming messages may allocate

    def max_new_voxels(self) -> int:
        """Upper bound of the voxels one exchange can allocate in this map (what `reserved` stands for)."""
        return len(self.sides) * self.rows_full

    def export(self):
        """Decide the kind of every message of this frame and enqueue the export kernels.  Afterwards: `out[name]` / `inp[name]` are the
        message buffers, `n_out[name]` / `n_in[name]` the rows (header included) that travel."""
        m, st = self.m, self.st
        f = self.f = st.setdefault("frame", 0)
        hist = st.setdefault("hist", {})
        self.note_out = self.note_in = None
        if self.on_gpu and self.sides:
            if "notes" not in st:           # four rotating slots of pinned header notes + the events that say when a slot's frame is done
                st["notes"] = [(torch.zeros((8,), dtype=torch.int32).pin_memory(), torch.zeros((8,), dtype=torch.int32).pin_memory()) for _ in range(4)]
                st["notes_np"] = [(a.numpy(), b.numpy()) for a, b in st["notes"]]
                st["events"] = [torch.cuda.Event() for _ in range(4)]
            self.note_out, self.note_in = st["notes"][f % 4]
            if f >= 4:
                hist[f - 4]["event"].synchronize()           # the slot's previous user
            for a in st["notes_np"][f % 4]:
                a[:] = 0
        kind = lambda which, k: _message_kind(st, f, which, k, self.rows_full, self.mode) if self.on_gpu else "full"
        self.kind_out = {name: kind("out", k) for name, _, k, _ in self.sides}
        self.kind_in = {name: kind("in", k) for name, _, k, _ in self.sides}
        if self.on_gpu and getattr(m, "_halo_lists_stale", False) and any(v == "delta" for v in self.kind_out.values()):
            raise RuntimeError("halo exchange: the map was changed by something other than integrate since the last exchange; call "
                               "parallel.restart_halo_exchange(buffers) on every rank")
        self.out, self.inp = {}, {}
        for name, _, _, _ in self.sides:
            if st.get("in_" + name) is None:
                st["in_" + name] = torch.zeros((1 + max(self.rows_full, 1), 32), dtype=torch.int32, device=self.dev)
            self.inp[name] = st["in_" + name]
        delta_sides = [name for name in self.kind_out if self.kind_out[name] == "delta"]
        if delta_sides:
            for name in delta_sides:
                if st.get("out_" + name) is None:
                    st["out_" + name] = torch.zeros((1 + max(self.rows_full, 1), 32), dtype=torch.int32, device=self.dev)
            m.export_halo_delta(st["out_left"] if "left" in delta_sides else None, st["out_right"] if "right" in delta_sides else None, self.note_out)
        full_hdr = {}
        for name, _, _, x0 in self.sides:
            if self.kind_out[name] == "full":
                st["out_" + name] = m.export_halo(x0, x0 + HALO, out=st.get("out_" + name))       # what that neighbour's halo mirrors
                full_hdr[name] = st["out_" + name]
            self.out[name] = st["out_" + name]
        if full_hdr and self.on_gpu:
            m.halo_lists_reset(full_hdr.get("left"), full_hdr.get("right"), self.note_out)
            if len(full_hdr) == len(self.sides):
                m._halo_lists_stale = False                  # every neighbour is being given the whole boundary
        self.n_out = {name: 1 + self.rows[self.kind_out[name]] for name in self.out}
        self.n_in = {name: 1 + self.rows[self.kind_in[name]] for name in self.inp}

    def transfer(self, group=None):
        """Both neighbours' messages over the process group: RCCL send/recv on device memory, or staged through pinned host memory when the
        group is gloo."""
        if not self.sides:
            return
        out, inp, n_out, n_in = self.out, self.inp, self.n_out, self.n_in
        if self.on_gpu and _host_staged(out[next(iter(out))], group):
            pins = self.st.setdefault("pins", {})
            host_out = {name: _pinned_like(out[name], pins, "o" + name) for name in out}
            host_in = {name: _pinned_like(inp[name], pins, "i" + name) for name in inp}
            for name in out:
                host_out[name][:n_out[name]].copy_(out[name][:n_out[name]], non_blocking=True)
            torch.cuda.current_stream().synchronize()
            ops = []
            for name, peer, _, _ in self.sides:
                ops.append(dist.P2POp(dist.isend, host_out[name][:n_out[name]], peer, group))
                ops.append(dist.P2POp(dist.irecv, host_in[name][:n_in[name]], peer, group))
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            for name in inp:
                inp[name][:n_in[name]].copy_(host_in[name][:n_in[name]], non_blocking=True)
            return
        ops = []
        for name, peer, _, _ in self.sides:
            ops.append(dist.P2POp(dist.isend, out[name][:n_out[name]], peer, group))
            ops.append(dist.P2POp(dist.irecv, inp[name][:n_in[name]], peer, group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()                      # RCCL: orders the current stream behind the transfer; the host does not block

    def transfer_loopback(self):
        """This process plays its slab alone: every message goes to itself through a device copy of the size that would go over the wire."""
        for name in self.inp:
            self.inp[name][:self.n_out[name]].copy_(self.out[name][:self.n_out[name]])

    def merge(self):
        m, st, f = self.m, self.st, self.f
        if self.sides and self.on_gpu:
            inp, rows = self.inp, self.rows
            m.merge_halo2(inp.get("left"), inp.get("right"), rows[self.kind_in["left"]] if "left" in inp else None,
                          rows[self.kind_in["right"]] if "right" in inp else None, self.note_in, reserved=self.reserved)
            ev = st["events"][f % 4]
            ev.record()
            no, ni = st["notes_np"][f % 4]
            st["hist"][f] = dict(event=ev, out=no, **{"in": ni}, kinds=(self.kind_out, self.kind_in),
                                 bytes_out=sum(self.n_out.values()) * 128, bytes_in=sum(self.n_in.values()) * 128)
            st["hist"].pop(f - 8, None)
        else:
            for name in self.inp:
                m.merge_halo(self.inp[name])
        st["frame"] = f + 1


def exchange_halo(m, rank: int, world: int, group=None, buffers: Optional[dict] = None, mode: str = "delta", loopback: bool = False,
                  reserved: bool = False):
    """Refresh the halo layers of `m` (a map with `set_ownership`) from the neighbouring slabs' owners: one send and one receive per
    neighbour (`ncclSend` / `ncclRecv` grouped by `batch_isend_irecv`; ring neighbours are directly xGMI-linked), nothing else.  Message
    sizes are known to the host (see above) and the record counts travel in the messages, so export, transfers and merge are all enqueued
    on the stream without a host round trip.  `buffers`: dict reused across frames (message tensors, history), filled on first use.
    mode: "delta" (bounded messages once the stream is in steady state) or "full" (whole boundary layers every frame).
    loopback: this process plays slab `rank` of `world` alone and hands every message to itself through a device copy of the size that
    would go over the wire (its own boundary records, merged with assign semantics: the map does not change) — export, transfer-sized
    copy and merge cost what they cost in the real exchange, which is what `bench.py --mode tiled --loopback S` times.
    With a gloo group and a map on the GPU the messages are staged through pinned host memory (two processes on one GPU, hosts without
    peer access); that path waits for the frame's integrate."""
    buffers = {} if buffers is None else buffers
    x = HaloExchange(m, rank, world, buffers, mode)
    x.reserved = reserved
    x.export()
    if loopback:
        x.transfer_loopback()
    else:
        x.transfer(group)
    x.merge()
    return buffers
