"""Synthetic-stream driver reproducing the per-frame cadence of the reference's `pytorch/main.py:refresh` (:42-102):
    pose @ points, pose.rotation @ normals  (main.py:83-84)  ->  map.integrate_keyframe (main.py:85)
    ->  map.extract_mesh(resolution, 4e6, max_std=0.15, interpolate=True)  (main.py:93)
with the north-star's harsher schedule (integrate AND mesh on every frame; the reference integrates 1 frame in 20).
Depth frames are rendered up front and stay resident in HBM; the timed part starts from depth + camera-frame normals."""
from __future__ import annotations

import ctypes
import struct
import time
from typing import List, Optional

import numpy as np
import torch

from . import _lib
from . import synthetic as syn
from .system.map import DenseIndexedMap


class FusionStream:
    def __init__(self, model, scene: syn.Scene, cfg: syn.MapConfig, intr: syn.Intrinsic, device: torch.device,
                 n_frames: int, deg_per_frame: float = 0.5, phase_deg: float = 0.0, orbit_radius: float = 0.3,
                 noise: bool = False, resolution: int = 4, max_n_triangles: int = int(4e6), max_std: float = 0.15,
                 initial_capacity: Optional[int] = None, tiling=None, halo_mode: str = "delta", halo_loopback: bool = False):
        """tiling = (rank, world, group): BASELINE config C5 — the grid is cut into `world` x-slabs, this stream's map owns slab `rank`,
        every rank is offered the whole frame and the 3 boundary layers are refreshed from the ring neighbours after every integrate
        (`parallel.exchange_halo`: one send + one receive per neighbour over RCCL / xGMI; halo_mode "delta": bounded messages with the
        frame's changes, "full": whole layers).  halo_loopback: this process plays slab `rank` of `world` alone and exchanges with
        itself (bench.py --loopback).  Eager / pipelined stepping only."""
        self.device = device
        self.intr = intr
        self.resolution, self.max_n_triangles, self.max_std = resolution, max_n_triangles, max_std
        if initial_capacity is None:
            # Room for the worst-case allocations of the frames the host keeps in flight.  A frame may allocate up to 7 voxels per
            # (prune_min_vox_obs + 1) points (a bound nothing comes near: a steady-state frame allocates tens), and the host's bound of
            # n_occupied lags the device by two frames; with less room than that, every frame would first have to wait for the previous
            # one to finish just to learn that there is space (`_ensure_capacity`), and the GPU would idle while the host enqueues.
            n_pts = intr.height * intr.width
            prune = int(cfg.namespace().prune_min_vox_obs)
            per_frame = 7 * (n_pts // (prune + 1)) if prune > 0 else 7 * n_pts
            if tiling is not None and tiling[1] > 1:
                per_frame += 2 * 3 * int(np.ceil((cfg.bound_max[1] - cfg.bound_min[1]) / cfg.voxel_size)) * int(np.ceil((cfg.bound_max[2] - cfg.bound_min[2]) / cfg.voxel_size))
            initial_capacity = 1 << 16
            while initial_capacity < 3 * per_frame + (1 << 16):
                initial_capacity *= 2
        self.map = DenseIndexedMap(model, cfg.namespace(), 29, device, initial_capacity=initial_capacity)
        self.tiling = tiling if (tiling is not None and tiling[1] > 1) else None
        self._halo_buffers = {}
        self.halo_mode, self.halo_loopback = halo_mode, bool(halo_loopback)
        if self.tiling is not None:
            from . import parallel
            rank, world, _ = self.tiling
            self.map.set_ownership(*parallel.slab_range(self.map.n_xyz[0], rank, world), halo=parallel.HALO)
        self.poses, self.depth, self.ncam = [], [], []
        for i in range(n_frames):
            R, t = syn.orbit_pose(i, orbit_radius, deg_per_frame, phase_deg)
            d, n = syn.render_frame(scene, R, t, intr, device, noise_seed=(1234 + i) if noise else None)
            self.poses.append(((ctypes.c_float * 9)(*[float(np.float32(v)) for v in R.reshape(-1)]),
                               (ctypes.c_float * 3)(*[float(np.float32(v)) for v in t])))
            self.depth.append(d)
            self.ncam.append(n)
        H, W = intr.height, intr.width
        self.xyz = torch.empty((H * W, 3), dtype=torch.float32, device=device)
        self.nrm = torch.empty((H * W, 3), dtype=torch.float32, device=device)
        self.stats = []
        self._pin = None
        self._pending = None
        self._copy_stream = torch.cuda.Stream(device=device)
        self._copy_done = None
        self._graphs = None
        self._graph_sig = None
        self._graph_export = True                                # captured graphs also hand each frame's new triangles to the host
        self.n_captures = 0
        self._g_in = None
        self._zc = None
        self._d_slots = None
        self._d_sig = None
        self._b_slots = None
        self._d2h_mode = "new"
        self.defer_export = True            # step_direct: a frame's new triangles travel to the host beside the NEXT frame's first kernel
        # direct / graph / batch / group frames: False = world points and normals are NOT written out per pixel (dif_integrate_frame with
        # xyz_world = normal_world = NULL: the stages that need a point recompute it from the depth pixel, bit-identically); True = into self.xyz / self.nrm
        self.keep_points = False
        self.backlog = []                   # outputs of a pending batch's earlier frames, when a frame-by-frame step had to complete it
        self.last_unq_mask = None           # eager / pipelined frames: the (H*W,) prune mask of the latest integrate (direct frames: `_d_mask`)
        # step_direct on TWO hardware queues (`enable_overlap`): frame i+1's integrate front end (unproject ... encoder) runs on `_fe_stream` beside
        # frame i's extract on the caller's stream, which carries fuse(i), extract(i), fuse(i+1), ... back to back: the front end waits — on the
        # device — for a word frame i's extract publishes when its fusion kernel is done, the fusion kernel of frame i+1 for a word the front end
        # publishes (dif_map_t.frame_seq / sync_words).  For d2h "dma" / "none" (an overlapped frame cannot carry the previous frame's deferred
        # export: that extract has not run yet) on an untiled map.
        self.overlap = False
        # step_direct: how many frames the host may have enqueued beyond the one it hands back.  1: step(i) returns frame i-1 (waits for it, exports
        # it, returns) before frame i+1 is enqueued — with two queues frame i+1's front end then reaches the GPU ~70 us after frame i-1's extract
        # ended, well into frame i's extract instead of at its start.  2: step(i) returns frame i-2, which is normally complete already: the host
        # runs ahead, the front end of the next frame is always queued in time (flush() returns what is left, oldest first, through `backlog`).
        self.host_depth = 1
        self._pending_older = None
        self._fe_stream = None
        self._fe_ptr = None
        self._mesh_stream = None            # where the frames' mesh halves run (a third hardware queue, or the front end's)
        self._mesh_ptr = None
        self._ov_active = False             # the frames in flight are overlapped ones (the two queues are coupled through the sync words)
        self._ov_seq = 0
        # ... and the frame's marching cubes + finish kernel beside the NEXT frame's decode (dif_extract_buffers_t.split_mesh): the extracts' stream
        # carries fuse, decode, fuse, decode, ...; frame n's mesh half is enqueued on the front-end stream behind frame n+1's integrate (whose fusion
        # kernel tells it that frame n's decode is done).  Consecutive frames alternate between two sets of extract buffers, batch maps, dirty totals
        # and counter blocks (slot parity).  Off by default: measured (profiles/r05_experiments.md 2) it is worth 2-3 % together with host_depth = 2
        # (7,090-7,290 against 6,910-7,050 frames/s) and nothing without, for twice the per-voxel extract buffers.
        self.split_mesh = False
        # ... and the frame's two extract scans (dirty-set compaction + neighbourhood marker, batch scan: two latency-bound launches, ~19 us) in its
        # FRONT END, before its fusion kernel (dif_map_t.scan_ahead): what a frame's extract decodes follows from what its encoder updated — the
        # extracts' stream then carries fuse, lattice decode, refine, marching cubes, finish only.  Needs the two buffer sets as well.  Off by default:
        # bit-identical, and measured (profiles/r05_experiments.md 2) 6,900-7,090 frames/s against 6,940-6,970 without, whatever the host depth — the
        # frame's ~170 us of launches take ~140 us on two queues however they are arranged.
        self.scan_ahead = False
        self._mesh_pending = None           # (frame number, slot, stamp) of the frame whose mesh half is not enqueued yet
        self.last_tensors = None            # the extract tensors of the frame enqueued last (tests)
        self.queues_independent = None      # what dif_queues_independent said about the two streams
        self._sdma = None                   # None: untried; True / False: the SDMA export works / does not (or is slow) in this process
        self._sdma_slow = 0
        self.sdma_us = []                   # (triangles, microseconds) of every SDMA export call

    def enable_overlap(self, on: bool = True) -> bool:
        """Two hardware queues for `step_direct`.  Returns whether the mode is on: it stays off (False) when no second stream on a hardware queue of
        its own can be had (HIP shares a queue between streams once more than GPU_MAX_HW_QUEUES are alive) — the frames would be serialised
        anyway and only pay for the device-side waits."""
        if not on or self.tiling is not None:
            self._ov_leave()
            self.overlap = False
            return False
        lib = _lib.load()
        with torch.cuda.device(self.device):
            main = _lib.stream_ptr()
            tried = []
            for _ in range(8):
                fe = torch.cuda.Stream(device=self.device)
                rc = int(lib.dif_queues_independent(main, ctypes.c_void_p(fe.cuda_stream)))
                if rc == 1:
                    self._fe_stream, self._fe_ptr, self.queues_independent, self.overlap = fe, ctypes.c_void_p(fe.cuda_stream), True, True
                    # a third queue for the frames' mesh halves (marching cubes + finish), independent of both; without one they share the front end's
                    self._mesh_stream, self._mesh_ptr = fe, self._fe_ptr
                    for _ in range(8):
                        ms = torch.cuda.Stream(device=self.device)
                        mp = ctypes.c_void_p(ms.cuda_stream)
                        if int(lib.dif_queues_independent(main, mp)) == 1 and int(lib.dif_queues_independent(self._fe_ptr, mp)) == 1:
                            self._mesh_stream, self._mesh_ptr = ms, mp
                            break
                        tried.append(ms)
                    return True
                tried.append(fe)            # (kept alive until the search ends, so that the pool hands out another one)
                if rc < 0:
                    break
            self.queues_independent = False
            self.overlap = False
            return False

    def _ov_enter(self):
        """First overlapped frame after anything else: the front-end stream starts behind everything the caller's stream holds, and the sync
        words (and their tickets) start from zero."""
        with torch.cuda.device(self.device):
            self.map._sync_words.zero_()
            self._ov_seq = 0
            self._fe_stream.wait_stream(torch.cuda.current_stream())
            self._mesh_stream.wait_stream(torch.cuda.current_stream())
        self._ov_active = True

    def _ov_leave(self):
        """Before anything but an overlapped frame touches the map on the caller's stream: the pending mesh half, then behind the front-end stream's
        last kernel."""
        if self._ov_active:
            self._ov_drain_mesh()
            with torch.cuda.device(self.device):
                torch.cuda.current_stream().wait_stream(self._fe_stream)
                torch.cuda.current_stream().wait_stream(self._mesh_stream)
            self._ov_active = False

    def _ov_frame_fields(self, seq: int, k: int, main_ptr):
        """dif_map_t fields of overlapped frame `seq` in slot k: its number, the extracts' stream, and the arrays of its parity."""
        m, cm, p = self.map, self.map._cmap, k & 1
        cm.frame_seq = seq
        cm.fuse_stream = main_ptr
        cm.dirty_tot = _lib.ptr(m._dirty_tot_b if p else m._dirty_tot)
        cm.vbm = _lib.ptr(m._vbm_b if p else m._vbm)
        cm.frame_counters = ctypes.c_void_p(m._frame_counters.data_ptr() + p * _lib.FC_COUNT * 4)
        # (mesh halves on a stream of their own: the front end of frame n waits for the mesh half of frame n-2, whose buffers frame n's decode reuses)
        cm.mesh_wait = max(0, seq - 2) if (self.split_mesh and self._mesh_stream is not self._fe_stream) else 0
        sa = bool(self.scan_ahead and self.resolution <= 4)
        cm.scan_ahead = 1 if sa else 0
        cm.front_stream = self._fe_ptr
        if sa:
            cm.grid_tot = _lib.ptr(m._grid_tot_b if p else m._grid_tot)

    def _ov_restore_fields(self):
        m, cm = self.map, self.map._cmap
        cm.frame_seq = 0
        cm.fuse_stream = None
        cm.frame_counters = None
        cm.mesh_wait = 0
        cm.scan_ahead = 0
        cm.front_stream = None
        cm.grid_tot = _lib.ptr(m._grid_tot)
        cm.dirty_tot = _lib.ptr(m._dirty_tot)
        cm.vbm = _lib.ptr(m._vbm)

    def _enqueue_mesh(self, stream_ptr, main_ptr, wait: int):
        """The mesh half (marching cubes + finish) of the frame in `_mesh_pending`, on `stream_ptr`."""
        seq, k, _ = self._mesh_pending
        self._mesh_pending = None
        self._ov_frame_fields(seq, k, main_ptr)
        try:
            _lib.check(self._d_lib.dif_extract_mesh(ctypes.byref(self.map._cmap), ctypes.byref(self._d_bufs[k]), int(self.resolution), float(self.max_std), 1,
                                                    int(wait), stream_ptr), "dif_extract_mesh")
        finally:
            self._ov_restore_fields()

    def _ov_drain_mesh(self):
        """A frame's mesh half that no later frame has carried onto the front-end stream yet (the last frame of a stream; before a safe point: log
        compaction, growth, a wait for that frame's stamp): on the caller's stream, behind its decode kernels and behind the front-end stream's earlier
        mesh halves (the log is appended to in frame order); later mesh halves on the mesh stream come behind it."""
        if self._mesh_pending is None:
            return
        with torch.cuda.device(self.device):
            main = torch.cuda.current_stream()
            main.wait_stream(self._mesh_stream)
            sp = _lib.stream_ptr()
            self._enqueue_mesh(sp, sp, 0)
            self._mesh_stream.wait_stream(main)

    def complete_frames(self):
        """Everything of the frames enqueued so far is on the GPU's queues (a split frame's mesh half included): a device synchronisation then leaves
        the map, the log and the counters in the state behind the last frame.  (Tests; the pipeline itself never needs it.)"""
        self._ov_drain_mesh()

    def _pts(self):
        return (_lib.ptr(self.xyz), _lib.ptr(self.nrm)) if self.keep_points else (_lib.ptr(None), _lib.ptr(None))

    def step(self, i: int, d2h: str = "new"):
        """One frame: unproject+transform (a1,a2) -> integrate (a3-a10) -> decode + marching cubes + mesh cache (a11-a16).
        d2h: "none" leaves the mesh in HBM; "new" copies this frame's new triangles to pinned host memory (async; written by the frame's
        kernels, or — direct frames — carried by the next frame's first kernels); "dma": the same delivery by hipMemcpyAsync on a side stream,
        once the frame's stamp has been seen (direct frames of one stream; elsewhere it delivers as "new");
        "full" copies the whole merged cache to the host like the reference's numpy cache."""
        intr = self.intr
        R, t = self.poses[i]
        self._ov_leave()
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().dif_unproject_transform(_lib.ptr(self.depth[i]), _lib.ptr(self.ncam[i]), _lib.ptr(self.xyz), _lib.ptr(self.nrm),
                                                           intr.height, intr.width, intr.fx, intr.fy, intr.cx, intr.cy, R, t, _lib.stream_ptr()),
                       "dif_unproject_transform")
        self.last_unq_mask = self.map.integrate_keyframe(self.xyz, self.nrm)
        self._exchange_halo()
        out = self.map.extract_mesh_arrays(self.resolution, self.max_n_triangles, max_std=self.max_std, to_host=(d2h == "full"))
        if d2h in ("new", "dma") and out is not None:
            tri, tid, tstd = self.map.mesh_cache_tensors(new_only=True)
            n = tri.size(0)
            if self._pin is None or self._pin[0].size(0) < n:
                cap = max(1 << 18, 2 * n)
                self._pin = (torch.empty((cap, 3, 3), dtype=torch.float32).pin_memory(), torch.empty((cap,), dtype=torch.long).pin_memory(),
                             torch.empty((cap, 3), dtype=torch.float32).pin_memory())
            self._pin[0][:n].copy_(tri, non_blocking=True)
            self._pin[1][:n].copy_(tid, non_blocking=True)
            self._pin[2][:n].copy_(tstd, non_blocking=True)
            out = (self._pin[0][:n], self._pin[1][:n], self._pin[2][:n])
        self.stats.append(dict(self.map.last_counters))
        return out

    def _exchange_halo(self, reserved: bool = False):
        if self.tiling is not None:
            from . import parallel
            rank, world, group = self.tiling
            with torch.cuda.device(self.device):
                parallel.exchange_halo(self.map, rank, world, group, self._halo_buffers, mode=self.halo_mode, loopback=self.halo_loopback,
                                       reserved=reserved)

    def _before_frame(self):
        """The D2H of the previous frame's triangles (side stream) reads a region of the mesh-cache LOG that later frames only append
        behind, so the next frame does not wait for it — unless the log is about to be compacted."""
        if self.map._gc_wanted:
            self._ov_drain_mesh()                         # (a pending mesh half appends to the log the compaction is about to move)
            self._complete_batch_before_gc(self._d2h_mode)
            self._export_deferred_now(self._pending)      # (the pending copy reads log positions that the compaction moves)
            if self._copy_done is not None:
                self._copy_done.synchronize()
                self._copy_done = None

    # ---- pipelined variant: no host wait inside the frame -----------------------------------------------------------------
    def _enqueue_frame(self, i: int):
        intr = self.intr
        R, t = self.poses[i]
        self._ov_leave()
        with torch.cuda.device(self.device):
            self._before_frame()
            _lib.check(_lib.load().dif_unproject_transform(_lib.ptr(self.depth[i]), _lib.ptr(self.ncam[i]), _lib.ptr(self.xyz), _lib.ptr(self.nrm),
                                                           intr.height, intr.width, intr.fx, intr.fy, intr.cx, intr.cy, R, t, _lib.stream_ptr()),
                       "dif_unproject_transform")
        self.last_unq_mask = self.map.integrate_keyframe(self.xyz, self.nrm)
        self._exchange_halo()
        return self.map.extract_mesh_enqueue(self.resolution, self.max_n_triangles, max_std=self.max_std)

    HOST_OUT_TRIANGLES = 1 << 18                                 # pinned staging per graph: 14 MB; larger updates fall back to _export_new
    SDMA_SLOW_US = 150.0

    def _export_new(self, handle, tri, tid, tstd):
        """Eager frames (and oversized updates): one small kernel on a side stream writes the three arrays straight into pinned host
        memory; it overlaps the next frame's kernels, which only READ this part of the log."""
        n = tri.size(0)
        if self._pin is None or self._pin[0].size(0) < n:
            cap = max(1 << 18, 2 * n)
            self._pin = (torch.empty((cap, 3, 3), dtype=torch.float32).pin_memory(), torch.empty((cap,), dtype=torch.long).pin_memory(),
                         torch.empty((cap, 3), dtype=torch.float32).pin_memory())
        with torch.cuda.device(self.device):
            if "event" in handle:            # (a stamped frame is complete by the time its handle gets here: extract_mesh_finish has seen the stamp)
                self._copy_stream.wait_event(handle["event"])
            with torch.cuda.stream(self._copy_stream):
                lo = tri.storage_offset() // 9
                b = self.map._cache_struct()
                _lib.check(_lib.load().dif_mesh_cache_export(ctypes.byref(b), lo, n, _lib.ptr(self._pin[0]), _lib.ptr(self._pin[1]),
                                                             _lib.ptr(self._pin[2]), _lib.stream_ptr()), "dif_mesh_cache_export")
                for src in (tri, tid, tstd):
                    src.record_stream(self._copy_stream)         # the log may be re-allocated (growth) while this is in flight
                self._copy_done = torch.cuda.Event()
                self._copy_done.record()
        return (self._pin[0][:n], self._pin[1][:n], self._pin[2][:n])

    def _export_deferred_now(self, handle):
        """A frame whose triangle export was deferred and that no later frame's first kernel has carried out yet: do the copy now
        (stream-ordered; `dif_export_pending`)."""
        if isinstance(handle, dict) and handle.get("deferred") and "export_event" not in handle and not handle.get("export_carried"):
            with torch.cuda.device(self.device):
                _lib.check(_lib.load().dif_export_pending(ctypes.byref(self.map._cmap), _lib.stream_ptr()), "dif_export_pending")
                ev = handle["host_slots"][handle["host_out"]]["export_event"]
                ev.record()
                handle["export_event"] = ev

    def _finish_frame(self, handle, d2h: str):
        if self._mesh_pending is not None and isinstance(handle, dict) and handle.get("stamp") == self._mesh_pending[2]:
            self._ov_drain_mesh()           # (its finish kernel, which stamps, is part of the mesh half)
        self._export_deferred_now(handle)
        tri, tid, tstd = self.map.extract_mesh_finish(handle)
        if handle.get("deferred"):
            if "export_event" in handle:
                handle["export_event"].synchronize()
            else:                           # carried out by the next frame's kernels, which stamp the slot's `notify` word when the copy is complete
                _lib.spin_until(handle["host_slots"][handle["host_out"]]["notify_np"], 0, handle["stamp"], "deferred triangle export")
        out = (tri, tid, tstd)
        if d2h == "new":
            n = tri.size(0)
            k = handle.get("host_out")
            if k is not None and n <= handle.get("host_capacity", self.HOST_OUT_TRIANGLES):
                hp = handle["host_slots"][k]["out"] if "host_slots" in handle else self._zc[2][k]      # already there: written by the frame's last kernel
                out = (hp[0][:n], hp[1][:n], hp[2][:n])
            else:
                out = self._export_new(handle, tri, tid, tstd)
        elif d2h == "dma":
            # The frame is complete — its stamp, written behind a system-scope fence by the last kernel of its extract, has been seen — and its new
            # triangles sit in the log (mesh left in HBM).  Three hipMemcpyAsync (blit kernels on this runtime) take them to the frame's pinned slot
            # on a side stream while the next frame's kernels, enqueued already, run: none of the frame's own kernels carries the PCIe transfer
            # and no event sits in the main queue.
            # (What the copy reads was written by an EARLIER kernel of the frame than the one that stamps: that kernel's end-of-kernel release
            # has written its lines back — on gfx942 / gfx950 every release at agent scope does, the eight L2s are not coherent with each other —
            # and the copy kernel starts with the matching acquire.)
            n = tri.size(0)
            k = handle.get("dma_slot")
            if n and k is not None and n <= self.HOST_OUT_TRIANGLES:
                sl = handle["host_slots"][k]
                if self._sdma is not False:
                    # by the SDMA engines themselves (HSA copy, three row ranges under one signal): no copy kernel on any queue; the call returns
                    # when the rows have landed.  A process whose HSA runtime cannot be reached falls back to hipMemcpyAsync (blit kernels), once —
                    # and so does one in which the engines are slow: ~35 us for a steady-state frame's rows normally, but 6 times that in a process
                    # that has returned gigabytes of device memory to the driver (torch.cuda.empty_cache(); bench.py:prime_process) — eight calls
                    # in a row above SDMA_SLOW_US with fewer than 2^15 triangles each switch it off.
                    t0 = time.perf_counter()
                    rc = _lib.load().dif_mesh_cache_export_sdma(ctypes.byref(self.map._cache_struct()), tri.storage_offset() // 9, n, sl["out_ptr"][0],
                                                                sl["out_ptr"][1], sl["out_ptr"][2])
                    self._sdma = (rc == 0)
                    self.sdma_us.append((int(n), (time.perf_counter() - t0) * 1e6))
                    if n < (1 << 15):
                        self._sdma_slow = self._sdma_slow + 1 if (time.perf_counter() - t0) * 1e6 > self.SDMA_SLOW_US else 0
                        if self._sdma_slow >= 8:
                            self._sdma = False
                if self._sdma:
                    hp = sl["out"]
                    self.stats.append(dict(self.map.last_counters))
                    return (hp[0][:n], hp[1][:n], hp[2][:n])
                with torch.cuda.device(self.device):
                    with torch.cuda.stream(self._copy_stream):
                        _lib.check(_lib.load().dif_mesh_cache_export_dma(ctypes.byref(self.map._cache_struct()), tri.storage_offset() // 9, n, sl["out_ptr"][0],
                                                                         sl["out_ptr"][1], sl["out_ptr"][2], _lib.stream_ptr()), "dif_mesh_cache_export_dma")
                        sl["export_event"].record()
                    sl["export_event"].synchronize()
                hp = sl["out"]
                out = (hp[0][:n], hp[1][:n], hp[2][:n])
            elif n:
                out = self._export_new(handle, tri, tid, tstd)
        elif d2h == "full":
            mc = self.map.mesh_cache
            out = (mc.vertices, mc.vertices_flatten_id, mc.vertices_std)
        self.stats.append(dict(self.map.last_counters))
        return out

    def step_pipelined(self, i: int, d2h: str = "new"):
        """Same work per frame as `step`, software-pipelined by one frame: frame i is enqueued, then frame i-1 (already finished or
        finishing on the GPU) is completed on the host — counter read-back, D2H of its new triangles on a side stream.  The mesh
        handed back is the previous frame's; call `flush()` after the last frame."""
        self._d2h_mode = d2h
        h = self._enqueue_frame(i)
        out = None
        done = self._finish_pending(d2h)                          # (a batch may be pending: its earlier frames go to `backlog`)
        if done:
            self.backlog += done[:-1]
            out = done[-1]
        self._pending = h
        return out

    def flush(self, d2h: str = "new"):
        """Complete what is pending; returns the last frame's output (`flush_all` returns every pending frame's; with host_depth 2 the frame
        before the last goes to `backlog`)."""
        outs = self.flush_all(d2h)
        if self.host_depth >= 2:
            self.backlog += outs[:-1]
        return outs[-1] if outs else None

    def flush_all(self, d2h: str = "new"):
        outs = self._finish_pending(d2h)
        self._ov_leave()
        with torch.cuda.device(self.device):
            self._copy_stream.synchronize()
        return outs

    # ---- direct variant: the frame's launches are enqueued by two C calls, the host stays ahead of the GPU ------------------------------
    # The same per-frame protocol as `step_graph` below (frame descriptor in, counters + new triangles out through pinned host memory,
    # written by the frame's first / last kernel; results picked up one frame later), but the 12 kernels are launched directly:
    # `dif_integrate_frame` + `dif_extract` enqueue them back to back in ~60 us of host time for ~190 us of GPU time, so the queue never
    # runs dry — provided the map has room for the worst-case allocations of the frames in flight (see __init__: otherwise every frame
    # first waits for its predecessor).  A replayed hipGraph has no kernel boundaries inside a frame, but consecutive graph launches on one
    # stream start ~33 us apart on the GPU (profiles/r02_timeline_graph.txt); this path is the faster one and needs no re-capture when a
    # buffer grows.  A frame's new triangles are copied to the host beside the NEXT frame's point kernels (`defer_export`).
    DIRECT_SLOTS = 4

    def _direct_prepare(self):
        m, dev = self.map, self.device
        H, W = self.intr.height, self.intr.width
        if self._d_slots is None:
            cap = self.HOST_OUT_TRIANGLES
            self._d_slots = [dict(frame=torch.zeros((64,), dtype=torch.uint8).pin_memory(),
                                  counters=torch.zeros((_lib.C_COUNT,), dtype=torch.int32).pin_memory(),
                                  out=(torch.empty((cap, 3, 3), dtype=torch.float32).pin_memory(), torch.empty((cap,), dtype=torch.long).pin_memory(),
                                       torch.empty((cap, 3), dtype=torch.float32).pin_memory()),
                                  notify=torch.zeros((2,), dtype=torch.int32).pin_memory(),
                                  event=torch.cuda.Event(), export_event=torch.cuda.Event()) for _ in range(self.DIRECT_SLOTS)]
            for sl in self._d_slots:
                sl["frame_np"] = sl["frame"].numpy()
                sl["counters_np"] = sl["counters"].numpy()
                sl["notify_np"] = sl["notify"].numpy()
                sl["out_ptr"] = tuple(_lib.ptr(t) for t in sl["out"])
            self._d_mask = torch.empty((H * W,), dtype=torch.uint8, device=dev)
            self._d_seq = 0
            self._d_desc = [np.frombuffer(struct.pack("<QQ12f", self.depth[i].data_ptr(), self.ncam[i].data_ptr(), *R, *t), dtype=np.uint8).copy()
                            for i, (R, t) in enumerate(self.poses)]
        two_sets = bool(self.overlap and (self.split_mesh or self.scan_ahead) and self.resolution <= 4)
        sig = (m._capacity, m._ws.data_ptr(), m._xbuf[0], m._cache[0].data_ptr(), self.HOST_OUT_TRIANGLES, two_sets)
        if self._d_sig != sig:                       # (re)build the per-slot buffer descriptors after a re-allocation
            # (a pending deferred export points into the mesh log: carry it out before anything below may re-allocate that log)
            self._ov_drain_mesh()
            self._export_deferred_now(self._pending)
            self._d_bufs, self._d_tens = [], []
            for k, sl in enumerate(self._d_slots):
                # (split frames: the slots of odd index use the second set of per-voxel buffers — frame n's marching cubes reads its cubes while
                # frame n+1's decode writes the other set's)
                tens, buf = m._extract_buffers(self.resolution, self.max_n_triangles, max_vox=self._stream_extract_voxels(), second=bool(two_sets and (k & 1)))
                buf.counters_out = _lib.ptr(sl["counters"])
                self._d_bufs.append(buf)
                self._d_tens.append(tens)
            self._d_sig = (m._capacity, m._ws.data_ptr(), m._xbuf[0], m._cache[0].data_ptr(), self.HOST_OUT_TRIANGLES, two_sets)
            self._d_w = m.model.packed.weights_struct(dev)
            self._d_lib = _lib.load()
            self._d_args = (self.intr.height, self.intr.width, self.intr.fx, self.intr.fy, self.intr.cx, self.intr.cy)

    def _stream_extract_voxels(self) -> int:
        """Rows of the per-voxel extract buffers of the direct / graph / batch paths.  Sized for the map's CAPACITY (a frame can then
        never overflow them and their addresses stay put while the occupancy grows) — ~7.7 KB per voxel at resolution 4, i.e. ~4 GB for
        the default 524,288-slot map of a 640x480 stream — but never beyond `map.extract_buffer_bytes` (default 8 GB; several streams
        on one GPU lower it): a frame that would need more rows raises the map's overflow error instead of tying up HBM for a bound
        (7 new voxels per 17 points) that no stream comes near."""
        m = self.map
        R = 2 * self.resolution
        per_voxel = (R ** 3) * 12 + (self.resolution ** 3) * 8 + 1024 + 64
        rows = m._capacity
        while rows > 4096 and rows * per_voxel > m.extract_buffer_bytes:
            rows //= 2
        return rows

    def _direct_begin(self, i: int, d2h: str):
        """Everything of a direct frame that precedes its launches: room in the map for the frames in flight, a due log compaction,
        the frame's pinned slot and buffer descriptor, the frame descriptor written.  To be called with the device current.  Returns
        (slot index, slot, extract buffers, export?, output of a frame that had to be completed early or None)."""
        m = self.map
        self._no_async_meshing()
        self._d2h_mode = d2h
        N = self.intr.height * self.intr.width
        prune = int(m.args.prune_min_vox_obs)
        may_add = 7 * (N // (prune + 1)) if prune > 0 else 7 * N
        if self.tiling is not None:         # the halo refresh between integrate and extract may allocate too: room for it is made up front, so that
            may_add += 2 * m.halo_message_rows(3)   # no buffer moves between the two C calls whose descriptors are prepared below
        out = None
        if m._ws is None or m._xbuf is None or m._cache is None:
            raise RuntimeError("run at least one eager step before step_direct (buffers are sized there)")
        if m._n_occ_ub + may_add > m._capacity:
            self._ov_drain_mesh()                            # (a growth re-allocates the batch maps a pending mesh half reads)
            if self._pending is not None:
                done = self._finish_pending(d2h)             # make the bound exact before deciding to grow
                self.backlog += done[:-1]
                out = done[-1]
        m._ensure_capacity(may_add)
        self._before_frame()
        if m._gc_wanted:
            self._complete_batch_before_gc(d2h)
            if self._pending_older is not None:           # (of two pending frames only the newer one's triangles are the compacted log's tail)
                q, self._pending_older = self._pending_older, None
                self.backlog.append(self._finish_frame(q, d2h))
            self._export_deferred_now(self._pending)      # (the pending copy reads log positions that the compaction moves)
            out = self._own_storage(out)
            self.backlog = [self._own_storage(o) for o in self.backlog]
            m._cache_gc()
        self._direct_prepare()
        k = self._d_seq % self.DIRECT_SLOTS
        self._d_seq += 1
        sl = self._d_slots[k]
        export = d2h == "new"
        self._stamp = (getattr(self, "_stamp", 0) % 0x3FFFFFFF) + 1
        buf = self._direct_fill(k, export)
        sl["frame_np"][:] = self._d_desc[i]
        return k, sl, buf, export, out

    def _direct_fill(self, k: int, export: bool):
        """The per-frame fields of slot k's extract-buffer descriptor (ONE place: `_direct_begin`, and again by `FusionStreamGroup.step` when
        another map's growth re-allocated this stream's buffers — and with them the descriptors — between begin and launch)."""
        sl, buf = self._d_slots[k], self._d_bufs[k]
        buf.out_tri, buf.out_id, buf.out_std = (_lib.ptr(t) for t in sl["out"]) if export else (None, None, None)
        buf.out_capacity = self.HOST_OUT_TRIANGLES if export else 0
        # Deferred export: this frame's extract leaves the copy of its new triangles (~0.5 MB over PCIe) to the FIRST kernel of the next
        # frame, where it overlaps the point pass instead of lengthening marching cubes; the host picks a frame's triangles up behind
        # that kernel (`export_event`), still before it enqueues the frame after.
        # (Only the one-pass marching cubes — resolution <= 4 — and the deferred copy write a frame's triangles BEFORE the kernel that
        # stamps; the two-pass kernels leave them to `k_extract_finish` itself, whose block 0 stamps without waiting for the other
        # workgroups' copies: there the export is always deferred, so that the stamp never stands for rows it does not cover.)
        buf.defer_export = 1 if (export and (self.defer_export or self.resolution > 4)) else 0
        # Completion without events: the extract's last kernel stamps the pinned counter snapshot, and the kernels that carry the deferred
        # copy out stamp `notify` — the host polls those words instead of waiting on events (an event record costs the queue ~5 us
        # between two kernels, twice per frame).  (Without a deferred export this frame's triangles are written by the one-pass
        # marching cubes, a kernel before the stamp.)
        buf.stamp = self._stamp
        buf.export_notify = _lib.ptr(sl["notify"]) if buf.defer_export else None
        buf.split_mesh = 0              # (set by the two-queue path of step_direct only)
        return buf

    def _direct_integrated(self):
        """Right behind the frame's integrate launches: the PREVIOUS frame's deferred triangle export has just been carried out by this
        frame's point kernels — record the event its host side waits for."""
        p = self._pending
        if isinstance(p, dict) and p.get("deferred") and "export_event" not in p:
            p["export_carried"] = True      # its fusion kernel stamps the slot's `notify` word: nothing to record

    def _direct_end(self, k, sl, buf, export, d2h, out):
        """Behind the frame's extract launches: the frame's handle; completes the previous frame on the host."""
        m = self.map
        m.mesh_cache.invalidate_host_copy()
        h = dict(stamp=int(buf.stamp), counters=sl["counters_np"], epoch=m._gc_epoch, add_total=m._add_total, max_n_triangles=self.max_n_triangles,
                 host_out=(k if export else None), host_slots=self._d_slots, deferred=bool(buf.defer_export))
        if d2h == "dma":
            h["dma_slot"] = k
        if self.host_depth >= 2 and (self._pending is None or (isinstance(self._pending, dict) and "stamp" in self._pending)):
            # two frames in flight: hand back the older one (normally complete by now), keep the previous frame pending
            if self._pending_older is not None:
                p, self._pending_older = self._pending_older, None
                done = self._finish_frame(p, d2h)
                if out is not None:
                    self.backlog.append(out)
                out = done
            self._pending_older, self._pending = self._pending, h
            if self.backlog:                                  # (frames a safe point had to complete early are older than anything else: oldest first)
                if out is not None:
                    self.backlog.append(out)
                out = self.backlog.pop(0)
            return out
        done = self._finish_pending(d2h)                          # (a batch may be pending: its earlier frames go to `backlog`)
        if done:
            self.backlog += done[:-1]
            out = done[-1]
        self._pending = h
        return out

    def step_direct(self, i: int, d2h: str = "new"):
        """One frame enqueued with two C calls (no graph), host one frame ahead; returns the previous frame's output like `step_pipelined`."""
        m = self.map
        with torch.cuda.device(self.device):
            ov = self.overlap and d2h in ("dma", "none")
            if not ov:
                self._ov_leave()
            k, sl, buf, export, out = self._direct_begin(i, d2h)
            lib, w, sp = self._d_lib, self._d_w, _lib.stream_ptr()
            H, W, fx, fy, cx, cy = self._d_args
            if ov:
                # two queues: this frame's front end on `_fe_stream` (behind the previous frame's fusion kernel, beside its extract), its fusion kernel
                # behind that extract, its extract — on the caller's stream — behind its fusion kernel: dif_map_t.frame_seq
                if not self._ov_active:
                    self._ov_enter()
                self._ov_seq += 1
                seq = self._ov_seq
                split = bool(self.split_mesh and self.resolution <= 4)
                buf.split_mesh = 1 if split else 0
                self.last_tensors = self._d_tens[k]
                try:
                    # (the fusion kernel goes to the extracts' stream — the caller's —, behind the previous frame's decode)
                    self._ov_frame_fields(seq, k, sp)
                    _lib.check(lib.dif_integrate_frame(ctypes.byref(m._cmap), ctypes.byref(w), _lib.ptr(sl["frame"]), H, W, fx, fy, cx, cy, *self._pts(),
                                                       _lib.ptr(self._d_mask), _lib.ptr(m._ws), m._ws.numel(), self._fe_ptr), "dif_integrate_frame")
                    self._direct_integrated()
                    if self._mesh_pending is not None:      # the previous frame's marching cubes + finish: the mesh stream, behind this frame's fusion kernel
                        self._enqueue_mesh(self._mesh_ptr, sp, 1)
                        self._ov_frame_fields(seq, k, sp)
                    _lib.check(lib.dif_extract(ctypes.byref(m._cmap), ctypes.byref(w), ctypes.byref(buf), int(self.resolution), 1, float(self.max_std), 0, 1, sp),
                               "dif_extract")
                    if split:
                        self._mesh_pending = (seq, k, int(buf.stamp))
                finally:
                    self._ov_restore_fields()
                return self._direct_end(k, sl, buf, export, d2h, out)
            buf.split_mesh = 0
            self.last_tensors = self._d_tens[k]
            _lib.check(lib.dif_integrate_frame(ctypes.byref(m._cmap), ctypes.byref(w), _lib.ptr(sl["frame"]), H, W, fx, fy, cx, cy, *self._pts(),
                                               _lib.ptr(self._d_mask), _lib.ptr(m._ws), m._ws.numel(), sp), "dif_integrate_frame")
            self._direct_integrated()
            if self.tiling is not None:
                self._exchange_halo(reserved=True)              # export -> send/recv -> merge, all on this stream, no host wait
            _lib.check(lib.dif_extract(ctypes.byref(m._cmap), ctypes.byref(w), ctypes.byref(buf), int(self.resolution), 1, float(self.max_std), 0, 1, sp),
                       "dif_extract")
            return self._direct_end(k, sl, buf, export, d2h, out)

    # ---- batched variant: F consecutive frames captured into ONE hipGraph -------------------------------------------------------------
    # Inside a replayed graph the kernels follow each other without the ~2 us boundary of separate launches, but consecutive graph
    # launches start ~33 us apart on the GPU (profiles/r02_timeline_graph.txt).  For a stream whose poses are known ahead (offline
    # reconstruction, this benchmark) F frames go into one graph: one such gap per F frames, no boundaries inside.  Every frame still
    # reads its own descriptor and hands its own counters / new triangles to pinned host memory; the host collects a batch's results
    # while the next batch runs.  Bit-identical to the frame-by-frame paths (tests/test_gpu_stream.py).
    BATCH_HOST_OUT_TRIANGLES = 1 << 16

    def _batch_prepare(self, F: int, export: bool):
        m, dev, intr = self.map, self.device, self.intr
        H, W = intr.height, intr.width
        lib = _lib.load()
        if self._b_slots is None or len(self._b_slots[0]) != F:
            cap = self.BATCH_HOST_OUT_TRIANGLES
            self._b_slots = [[dict(frame=torch.zeros((64,), dtype=torch.uint8).pin_memory(),
                                   counters=torch.zeros((_lib.C_COUNT,), dtype=torch.int32).pin_memory(),
                                   out=(torch.empty((cap, 3, 3), dtype=torch.float32).pin_memory(), torch.empty((cap,), dtype=torch.long).pin_memory(),
                                        torch.empty((cap, 3), dtype=torch.float32).pin_memory())) for _ in range(F)] for _ in range(2)]
            for g in self._b_slots:
                for sl in g:
                    sl["frame_np"], sl["counters_np"] = sl["frame"].numpy(), sl["counters"].numpy()
            self._b_events = [torch.cuda.Event(), torch.cuda.Event()]
            self._b_mask = torch.empty((H * W,), dtype=torch.uint8, device=dev)
            self._b_seq = 0
            self._b_sig = None
            if self._d_slots is None:
                self._d_desc = None
            self._b_desc = [np.frombuffer(struct.pack("<QQ12f", self.depth[i].data_ptr(), self.ncam[i].data_ptr(), *R, *t), dtype=np.uint8).copy()
                            for i, (R, t) in enumerate(self.poses)]
        sig = (export, F, m._capacity, m._ws.data_ptr(), m._xbuf[0], m._cache[0].data_ptr())
        if self._b_sig != sig:
            torch.cuda.synchronize()
            w = m.model.packed.weights_struct(dev)
            self._b_graphs = []
            for g in range(2):
                bufs = []
                for sl in self._b_slots[g]:
                    _, buf = m._extract_buffers(self.resolution, self.max_n_triangles, max_vox=self._stream_extract_voxels())
                    buf.counters_out = _lib.ptr(sl["counters"])
                    if export:
                        buf.out_tri, buf.out_id, buf.out_std = (_lib.ptr(t) for t in sl["out"])
                        buf.out_capacity = self.BATCH_HOST_OUT_TRIANGLES
                    bufs.append(buf)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    sp = _lib.stream_ptr()
                    for sl, buf in zip(self._b_slots[g], bufs):
                        _lib.check(lib.dif_integrate_frame(ctypes.byref(m._cmap), ctypes.byref(w), _lib.ptr(sl["frame"]), H, W, intr.fx, intr.fy, intr.cx,
                                                           intr.cy, *self._pts(), _lib.ptr(self._b_mask), _lib.ptr(m._ws),
                                                           m._ws.numel(), sp), "dif_integrate_frame")
                        _lib.check(lib.dif_extract(ctypes.byref(m._cmap), ctypes.byref(w), ctypes.byref(buf), int(self.resolution), 1,
                                                   float(self.max_std), 0, 1, sp), "dif_extract")
                self._b_graphs.append((graph, bufs))
            self._b_sig = (export, F, m._capacity, m._ws.data_ptr(), m._xbuf[0], m._cache[0].data_ptr())
            self.n_captures += 1

    def _no_async_meshing(self):
        """The direct / graph / batch paths enqueue integrate and extract on ONE stream without the map's lock, its `_integrate_done` event
        or a wait on `meshing_stream`: they are exclusive with `extract_mesh(extract_async=True)` on the same map."""
        if self.map.meshing_thread is not None and self.map.meshing_thread.is_alive():
            raise RuntimeError("step_direct / step_graph / step_batch cannot be mixed with an asynchronous extract_mesh on the same map")

    def step_batch(self, i0: int, F: int, d2h: str = "new"):
        """Frames i0 .. i0+F-1 enqueued as ONE graph launch; returns the list of the outputs of everything that was pending before
        (the previous batch, a frame of one of the frame-by-frame paths, and anything an earlier call left in `backlog`), oldest first."""
        m = self.map
        self._no_async_meshing()
        self._ov_leave()
        self._d2h_mode = d2h
        if self.tiling is not None:
            raise RuntimeError("the spatially tiled stream exchanges halos between kernels of a frame: step / step_pipelined only")
        N = self.intr.height * self.intr.width
        prune = int(m.args.prune_min_vox_obs)
        may_add = F * (7 * (N // (prune + 1)) if prune > 0 else 7 * N)
        outs = []
        with torch.cuda.device(self.device):
            if m._ws is None or m._xbuf is None or m._cache is None:
                raise RuntimeError("run at least one eager step before step_batch (buffers are sized there)")
            if m._n_occ_ub + may_add > m._capacity:
                outs += self._finish_pending(d2h)                # make the bound exact before deciding to grow
            m._ensure_capacity(may_add)
            self._before_frame()
            if m._gc_wanted:
                outs += self._finish_pending(d2h)
                outs = [self._own_storage(o) for o in outs]
                self.backlog = [self._own_storage(o) for o in self.backlog]
                m._cache_gc()
            export = d2h == "new"
            self._batch_prepare(F, export)
            g = self._b_seq % 2
            self._b_seq += 1
            for j, sl in enumerate(self._b_slots[g]):
                sl["frame_np"][:] = self._b_desc[i0 + j]
            self._b_graphs[g][0].replay()
            m.mesh_cache.invalidate_host_copy()
            ev = self._b_events[g]
            ev.record()
            hs = [dict(event=ev, counters=sl["counters_np"], epoch=m._gc_epoch, add_total=m._add_total, max_n_triangles=self.max_n_triangles,
                       host_out=(j if export else None), host_slots=self._b_slots[g], host_capacity=self.BATCH_HOST_OUT_TRIANGLES)
                  for j, sl in enumerate(self._b_slots[g])]
        outs += self._finish_pending(d2h)
        self._pending = hs
        # whatever a log compaction above had to complete early went to `backlog`: it is older than everything in `outs`
        outs, self.backlog = self.backlog + outs, []
        return outs

    @staticmethod
    def _own_storage(o):
        """An output handed back as DEVICE views of the mesh-cache log (d2h "none") gets storage of its own: a log compaction moves the rows the
        views name.  (A frame completed early in the very call that then compacts the log — a map short of room — came back with rows of
        other triangles until round 5.)"""
        return o if (o is None or not o[0].is_cuda) else tuple(x.clone() for x in o)

    def _complete_batch_before_gc(self, d2h: str):
        """A compaction of the mesh-cache log moves entries: with ONE extract pending its triangles are simply the new log's tail, with a
        batch pending they are not — so a pending batch is completed (into `backlog`) before the log is compacted."""
        if isinstance(self._pending, list):
            self.backlog += self._finish_pending(d2h)

    def _finish_pending(self, d2h: str):
        """Complete whatever is pending (one frame handle or a batch's list of them; with host_depth 2 the frame before it first); returns the
        outputs, oldest first."""
        outs = []
        if self._pending_older is not None:
            q, self._pending_older = self._pending_older, None
            outs.append(self._finish_frame(q, d2h))
        p, self._pending = self._pending, None
        if p is None:
            return outs
        if not isinstance(p, list):
            return outs + [self._finish_frame(p, d2h)]
        for h in p:
            out = self._finish_frame(h, d2h)
            if self._pin is not None and out[0].numel() and out[0].data_ptr() == self._pin[0].data_ptr():
                # an update larger than the batch's pinned staging went through the shared fallback buffer: the next frame of the batch
                # will reuse it, so this one gets a copy of its own
                self._copy_done.synchronize()
                out = tuple(x.clone() for x in out)
            outs.append(out)
        return outs

    # ---- hipGraph variant: the 14 launches of a frame are captured once and replayed ----------------------------------------
    # Everything a frame launches has host-independent shapes (device counters carry the sizes), so a frame is a static graph: write the
    # frame descriptor (input pointers + pose) into pinned host memory, replay, read the counters and the new triangles out of pinned
    # host memory one frame later.  Re-captured only when a buffer is re-allocated (capacity growth).
    def _graph_signature(self):
        m = self.map
        return (self._graph_export, m._capacity, m._ws.data_ptr() if m._ws is not None else 0, m._xbuf[0] if m._xbuf else None,
                m._cache[0].data_ptr() if m._cache else 0)

    def _capture_graphs(self):
        m, intr, dev = self.map, self.intr, self.device
        lib = _lib.load()
        H, W = intr.height, intr.width
        N = H * W
        with torch.cuda.device(dev):
            if self._g_in is None:
                self._g_in = (torch.zeros((64,), dtype=torch.uint8, device=dev), torch.empty((N,), dtype=torch.uint8, device=dev))
                self._g_frame_host = [torch.zeros((64,), dtype=torch.uint8).pin_memory() for _ in range(4)]
                self._g_counters = [torch.empty((_lib.C_COUNT,), dtype=torch.int32).pin_memory() for _ in range(4)]
                self._g_seq = 0
            if self._zc is None:
                # Two captured graphs in ping-pong, each bound to its own pinned HOST slots: the first kernel reads the frame descriptor
                # (dif_frame_t: two device pointers + pose, 64 bytes) straight out of host memory and the last kernel writes the
                # counters into host memory — no copy kernels on the launch stream.  A slot is reused two frames later, after the host
                # has finished the frame that used it.
                cap = self.HOST_OUT_TRIANGLES
                self._zc = ([torch.zeros((64,), dtype=torch.uint8).pin_memory() for _ in range(2)],
                            [torch.zeros((_lib.C_COUNT,), dtype=torch.int32).pin_memory() for _ in range(2)],
                            [(torch.empty((cap, 3, 3), dtype=torch.float32).pin_memory(), torch.empty((cap,), dtype=torch.long).pin_memory(),
                              torch.empty((cap, 3), dtype=torch.float32).pin_memory()) for _ in range(2)])
            _, mask = self._g_in
            w = m.model.packed.weights_struct(dev)
            torch.cuda.synchronize()
            graphs = []
            for k in range(2):
                _, buf = m._extract_buffers(self.resolution, self.max_n_triangles, max_vox=self._stream_extract_voxels())
                buf.counters_out = _lib.ptr(self._zc[1][k])
                if self._graph_export:                           # the frame's last kernel also writes its new triangles to pinned host memory
                    buf.out_tri, buf.out_id, buf.out_std = (_lib.ptr(t) for t in self._zc[2][k])
                    buf.out_capacity = self.HOST_OUT_TRIANGLES
                frame = self._zc[0][k]
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    sp = _lib.stream_ptr()
                    _lib.check(lib.dif_integrate_frame(ctypes.byref(m._cmap), ctypes.byref(w), _lib.ptr(frame), H, W, intr.fx, intr.fy, intr.cx, intr.cy,
                                                       *self._pts(), _lib.ptr(mask), _lib.ptr(m._ws), m._ws.numel(), sp),
                               "dif_integrate_frame")
                    _lib.check(lib.dif_extract(ctypes.byref(m._cmap), ctypes.byref(w), ctypes.byref(buf), int(self.resolution), 1, float(self.max_std),
                                               0, 1, sp), "dif_extract")
                graphs.append((g, buf))
            self._graphs = graphs
            self._graph_sig = self._graph_signature()
            self.n_captures += 1

    def step_graph(self, i: int, d2h: str = "new"):
        """`step_pipelined` with the frame's launches replayed from a captured hipGraph (host cost: writing a 64-byte frame descriptor
        into pinned memory and one graph launch)."""
        m = self.map
        self._no_async_meshing()
        self._ov_leave()
        self._d2h_mode = d2h
        if self.tiling is not None:
            raise RuntimeError("the spatially tiled stream exchanges halos between kernels of a frame: step / step_pipelined only")
        N = self.intr.height * self.intr.width
        prune = int(m.args.prune_min_vox_obs)
        may_add = 7 * (N // (prune + 1)) if prune > 0 else 7 * N
        out = None
        with torch.cuda.device(self.device):
            if m._ws is None or m._xbuf is None or m._cache is None:
                raise RuntimeError("run at least one eager step before step_graph (buffers are sized there)")
            if m._n_occ_ub + may_add > m._capacity:
                if self._pending is not None:                    # make the bound exact before deciding to grow
                    done = self._finish_pending(d2h)
                    self.backlog += done[:-1]
                    out = done[-1]
            m._ensure_capacity(may_add)
            self._before_frame()
            if m._gc_wanted:
                self._complete_batch_before_gc(d2h)
                out = self._own_storage(out)
                self.backlog = [self._own_storage(o) for o in self.backlog]
                m._cache_gc()
            self._graph_export = (d2h == "new")
            if self._graphs is None or self._graph_sig != self._graph_signature():
                torch.cuda.synchronize()
                self._capture_graphs()
            k = self._g_seq % 2
            self._g_seq += 1
            R, t = self.poses[i]
            self._zc[0][k].numpy()[:] = np.frombuffer(struct.pack("<QQ12f", self.depth[i].data_ptr(), self.ncam[i].data_ptr(), *R, *t), dtype=np.uint8)
            self._graphs[k][0].replay()
            m.mesh_cache.invalidate_host_copy()
            pc = self._zc[1][k]                                    # written by the frame's last kernel; read after the event
            ev = torch.cuda.Event()
            ev.record()
            h = dict(event=ev, counters=pc, epoch=m._gc_epoch, add_total=m._add_total, max_n_triangles=self.max_n_triangles,
                     host_out=(k if self._graph_export else None))
        done = self._finish_pending(d2h)                          # (a batch may be pending: its earlier frames go to `backlog`)
        if done:
            self.backlog += done[:-1]
            out = done[-1]
        self._pending = h
        return out


class FusionStreamGroup:
    """S independent subsequences on ONE GPU whose frames share their twelve launches (`dif_integrate_frames` + `dif_extract_streams`):
    BASELINE config C4's unit of work — an independent stream with a private map — batched inside a GPU.  A steady-state frame of one
    stream is 1-2 MLP tiles per SIMD and nine launches on their latency floors; S of them per launch fill the SIMDs and pay each floor
    once.  Every stream keeps its own map, pinned slots, deferred export and host bookkeeping (`FusionStream`), and its results are
    bit-identical to stepping it alone (`tests/test_gpu_stream.py::test_stream_group_matches_single_streams`)."""

    def __init__(self, streams: List[FusionStream]):
        if not 1 <= len(streams) <= _lib.MAX_STREAMS:
            raise ValueError(f"1..{_lib.MAX_STREAMS} streams per group")
        a = streams[0]
        for st in streams:
            if st.tiling is not None:
                raise ValueError("spatially tiled streams exchange halos inside a frame: not groupable")
            if st.device != a.device or (st.intr.height, st.intr.width) != (a.intr.height, a.intr.width) or st.map.n_xyz != a.map.n_xyz \
                    or st.resolution != a.resolution or st.max_std != a.max_std or st.map.model is not a.map.model:
                raise ValueError("the streams of a group share device, network, frame size, grid shape and extract parameters")
        self.streams = list(streams)
        self.device = a.device
        self._frames = (_lib.DifStreamFrame * len(streams))()

    MIN_CAPACITY = 8192         # dif_extract_streams' dirty-set scan walks whole 256-slot blocks of maps with more than 4,096 slots

    def _equalise_capacity(self):
        """The batched launches are shaped by ONE capacity: a map that has grown pulls the others along (rare: the capacity covers three
        frames of worst-case allocations, see FusionStream.__init__)."""
        cap = max(self.MIN_CAPACITY, max(st.map._capacity for st in self.streams))
        for st in self.streams:
            if st.map._capacity != cap:
                st._export_deferred_now(st._pending)
                with st.map._state_lock:
                    st.map._alloc_state(cap)

    def step(self, idx, d2h: str = "new"):
        """Frame idx[j] (or the same index for all, if an int) of every stream j, enqueued with two C calls for the whole group; returns the
        list of the streams' previous-frame outputs (None entries on the first call), like `FusionStream.step_direct`."""
        S = len(self.streams)
        idx = [idx] * S if isinstance(idx, int) else list(idx)
        a = self.streams[0]
        with torch.cuda.device(self.device):
            begun = [st._direct_begin(i, d2h) for st, i in zip(self.streams, idx)]
            caps = {st.map._capacity for st in self.streams}
            if len(caps) > 1 or min(caps) < self.MIN_CAPACITY:
                self._equalise_capacity()
                for st in self.streams:
                    st._direct_prepare()                       # descriptors of re-allocated buffers
                # (descriptors rebuilt by _direct_prepare are blank: the frame's stamp, notify word and output fields go in again)
                begun = [(k, sl, st._direct_fill(k, export), export, out) for st, (k, sl, _, export, out) in zip(self.streams, begun)]
            rows = {int(buf.max_voxels) for (_, _, buf, _, _) in begun}
            if len(rows) > 1:
                raise ValueError(f"the streams of a group need extract buffers of one size (rows per stream: {sorted(rows)}): give their maps the same "
                                 "`extract_buffer_bytes`")
            for j, (st, (k, sl, buf, export, out)) in enumerate(zip(self.streams, begun)):
                f, m = self._frames[j], st.map
                f.map = ctypes.pointer(m._cmap)
                f.frame_dev = _lib.ptr(sl["frame"])
                f.xyz_world, f.normal_world = st._pts()
                f.unq_mask = _lib.ptr(st._d_mask)
                f.ws, f.ws_bytes = _lib.ptr(m._ws), m._ws.numel()
                f.buf = ctypes.pointer(buf)
            lib, w, sp = a._d_lib, a._d_w, _lib.stream_ptr()
            H, W, fx, fy, cx, cy = a._d_args
            for st in self.streams:
                st._ov_leave()                      # (S streams per launch fill the machine as they are: a group's frames stay on one queue)
            _lib.check(lib.dif_integrate_frames(self._frames, S, ctypes.byref(w), H, W, fx, fy, cx, cy, sp), "dif_integrate_frames")
            for st in self.streams:
                st._direct_integrated()
            _lib.check(lib.dif_extract_streams(self._frames, S, ctypes.byref(w), int(a.resolution), float(a.max_std), 1, sp), "dif_extract_streams")
            return [st._direct_end(k, sl, buf, export, d2h, out) for st, (k, sl, buf, export, out) in zip(self.streams, begun)]

    def flush(self, d2h: str = "new"):
        return [st.flush(d2h) for st in self.streams]
