"""Synthetic-stream driver reproducing the per-frame cadence of the reference's `pytorch/main.py:refresh` (:42-102):
    pose @ points, pose.rotation @ normals  (main.py:83-84)  ->  map.integrate_keyframe (main.py:85)
    ->  map.extract_mesh(resolution, 4e6, max_std=0.15, interpolate=True)  (main.py:93)
with the north-star's harsher schedule (integrate AND mesh on every frame; the reference integrates 1 frame in 20).
Depth frames are rendered up front and stay resident in HBM; the timed part starts from depth + camera-frame normals.

Three ways to step a frame, one result (tests/test_gpu_stream.py):
  step / step_pipelined  the façade's own calls (`DenseIndexedMap.integrate_keyframe` + `extract_mesh_*`): they size every buffer;
  step_direct            the frame's launches enqueued by two C calls, the host one frame ahead, on one or two hardware queues;
  FusionStreamGroup      S streams whose frames share their launches.
(Rounds 2-5 also carried a captured-hipGraph variant, F frames per graph, the frame's mesh half on a third queue and the extract scans in
the front end: all measured slower than or equal to `step_direct` — profiles/r05_experiments.md — and removed in round 6.)"""
from __future__ import annotations

import ctypes
import struct
import time
from typing import List, Optional

import numpy as np
import torch

from . import _lib
from . import synthetic as syn
from .system.map import DenseIndexedMap, _next_pow2


class FusionStream:
    HOST_OUT_TRIANGLES = 1 << 18        # pinned staging per slot: 14 MB; larger updates fall back to _export_new
    SDMA_SLOW_US = 150.0
    DIRECT_SLOTS = 4

    def __init__(self, model, scene: syn.Scene, cfg: syn.MapConfig, intr: syn.Intrinsic, device: torch.device,
                 n_frames: int, deg_per_frame: float = 0.5, phase_deg: float = 0.0, orbit_radius: float = 0.3,
                 noise: bool = False, resolution: int = 4, max_n_triangles: int = int(4e6), max_std: float = 0.15,
                 initial_capacity: Optional[int] = None, tiling=None, halo_mode: str = "delta", halo_loopback: bool = False):
        """tiling = (rank, world, group): BASELINE config C5 — the grid is cut into `world` x-slabs, this stream's map owns slab `rank`,
        every rank is offered the whole frame and the 3 boundary layers are refreshed from the ring neighbours after every integrate
        (`parallel.exchange_halo`: one send + one receive per neighbour over RCCL / xGMI; halo_mode "delta": bounded messages with the
        frame's changes, "full": whole layers).  halo_loopback: this process plays slab `rank` of `world` alone and exchanges with
        itself (bench.py --loopback)."""
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
        self._d_slots = None
        self._d_sig = None
        self._d2h_mode = "new"
        self._stamp = 0
        self.defer_export = True            # step_direct: a frame's new triangles travel to the host beside the NEXT frame's first kernel
        # direct / group frames: False = world points and normals are NOT written out per pixel (dif_integrate_frame with xyz_world =
        # normal_world = NULL: the stages that need a point recompute it from the depth pixel, bit-identically); True = into self.xyz / self.nrm
        self.keep_points = False
        self.last_unq_mask = None           # eager / pipelined frames: the (H*W,) prune mask of the latest integrate (direct frames: `_d_mask`)
        # step_direct on TWO hardware queues (`enable_overlap`): frame i+1's integrate front end (unproject ... encoder) runs on `_fe_stream` beside
        # frame i's extract on the caller's stream, which carries fuse(i), extract(i), fuse(i+1), ... back to back: the front end waits — on the
        # device — for a word frame i's extract publishes when its fusion kernel is done, the fusion kernel of frame i+1 for a word the front end
        # publishes (dif_map_t.frame_seq / sync_words).  For d2h "dma" / "none" (an overlapped frame cannot carry the previous frame's deferred
        # export: that extract has not run yet) on an untiled map at resolution <= 4.
        self.overlap = False
        self._fe_stream = None
        self._fe_ptr = None
        self._ov_main = None                # the stream `enable_overlap` tested the front-end stream against (raw pointer value)
        self._ov_active = False             # the frames in flight are overlapped ones (the two queues are coupled through the sync words)
        self._ov_seq = 0
        self.last_tensors = None            # the extract tensors of the frame enqueued last (tests)
        self.queues_independent = None      # what dif_queues_independent said about the two streams
        self.sdma = None                    # None: untried; True / False: the SDMA export works / does not (or is slow) in this process
        self.force_export = None            # tests: "blit" = hipMemcpyAsync fallback, "kernel" = dif_mesh_cache_export, None = SDMA first
        self._sdma_slow = 0
        self.sdma_us = []                   # (triangles, microseconds) of every SDMA export call

    # ---- two hardware queues -----------------------------------------------------------------------------------------------------------
    def enable_overlap(self, on: bool = True) -> bool:
        """Two hardware queues for `step_direct`.  Returns whether the mode is on: it stays off (False) when no second stream on a hardware queue of
        its own can be had (HIP shares a queue between streams once more than GPU_MAX_HW_QUEUES are alive) — the frames would be serialised
        anyway and only pay for the device-side waits.  The check holds for the stream that is current NOW: `step_direct` on another stream
        runs on one queue."""
        self._ov_leave()
        self.overlap = False
        if not on or self.tiling is not None:
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
                    self._ov_main = main.value
                    return True
                tried.append(fe)            # (kept alive until the search ends, so that the pool hands out another one)
                if rc < 0:
                    break
            self.queues_independent = False
            return False

    def _ov_enter(self):
        """First overlapped frame after anything else: the front-end stream starts behind everything the caller's stream holds, and the sync
        words (and their tickets) start from zero."""
        with torch.cuda.device(self.device):
            self.map._sync_words.zero_()
            self._ov_seq = 0
            self._fe_stream.wait_stream(torch.cuda.current_stream())
        self._ov_active = True

    def _ov_leave(self):
        """Before anything but an overlapped frame touches the map on the caller's stream (an eager frame, a re-allocation, a log compaction):
        behind the front-end stream's last kernel."""
        if self._ov_active:
            with torch.cuda.device(self.device):
                torch.cuda.current_stream().wait_stream(self._fe_stream)
            self._ov_active = False

    def _pts(self):
        return (_lib.ptr(self.xyz), _lib.ptr(self.nrm)) if self.keep_points else (_lib.ptr(None), _lib.ptr(None))

    # ---- eager frames: the façade's own calls --------------------------------------------------------------------------------------------
    def _unproject(self, i: int):
        intr = self.intr
        R, t = self.poses[i]
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().dif_unproject_transform(_lib.ptr(self.depth[i]), _lib.ptr(self.ncam[i]), _lib.ptr(self.xyz), _lib.ptr(self.nrm),
                                                           intr.height, intr.width, intr.fx, intr.fy, intr.cx, intr.cy, R, t, _lib.stream_ptr()),
                       "dif_unproject_transform")

    def step(self, i: int, d2h: str = "new"):
        """One frame: unproject+transform (a1,a2) -> integrate (a3-a10) -> decode + marching cubes + mesh cache (a11-a16).
        d2h: "none" leaves the mesh in HBM; "new" copies this frame's new triangles to pinned host memory (async; written by the frame's
        kernels, or — direct frames — carried by the next frame's first kernels); "dma": the same delivery by the copy engines once the
        frame's stamp has been seen (direct frames of one stream; elsewhere it delivers as "new");
        "full" copies the whole merged cache to the host like the reference's numpy cache."""
        self._ov_leave()
        self._unproject(i)
        self.last_unq_mask = self.map.integrate_keyframe(self.xyz, self.nrm)
        self._exchange_halo()
        out = self.map.extract_mesh_arrays(self.resolution, self.max_n_triangles, max_std=self.max_std, to_host=(d2h == "full"))
        if d2h in ("new", "dma") and out is not None:
            tri, tid, tstd = self.map.mesh_cache_tensors(new_only=True)
            n = tri.size(0)
            self._ensure_pin(n)
            self._pin[0][:n].copy_(tri, non_blocking=True)
            self._pin[1][:n].copy_(tid, non_blocking=True)
            self._pin[2][:n].copy_(tstd, non_blocking=True)
            out = (self._pin[0][:n], self._pin[1][:n], self._pin[2][:n])
        self.stats.append(dict(self.map.last_counters))
        return out

    def _ensure_pin(self, n: int):
        if self._pin is None or self._pin[0].size(0) < n:
            cap = max(1 << 18, 2 * n)
            self._pin = (torch.empty((cap, 3, 3), dtype=torch.float32).pin_memory(), torch.empty((cap,), dtype=torch.long).pin_memory(),
                         torch.empty((cap, 3), dtype=torch.float32).pin_memory())

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
            self._ov_leave()
            self._export_deferred_now(self._pending)      # (the pending copy reads log positions that the compaction moves)
            if self._copy_done is not None:
                self._copy_done.synchronize()
                self._copy_done = None

    def step_pipelined(self, i: int, d2h: str = "new"):
        """Same work per frame as `step`, software-pipelined by one frame: frame i is enqueued, then frame i-1 (already finished or
        finishing on the GPU) is completed on the host — counter read-back, D2H of its new triangles on a side stream.  The mesh
        handed back is the previous frame's; call `flush()` after the last frame."""
        self._d2h_mode = d2h
        self._ov_leave()
        with torch.cuda.device(self.device):
            self._before_frame()
        self._unproject(i)
        self.last_unq_mask = self.map.integrate_keyframe(self.xyz, self.nrm)
        self._exchange_halo()
        h = self.map.extract_mesh_enqueue(self.resolution, self.max_n_triangles, max_std=self.max_std)
        out = self._finish_pending(d2h)
        self._pending = h
        return out

    def flush(self, d2h: str = "new"):
        """Complete what is pending; returns that frame's output (None if nothing was)."""
        out = self._finish_pending(d2h)
        self._ov_leave()
        with torch.cuda.device(self.device):
            self._copy_stream.synchronize()
        return out

    # ---- delivery of a frame's new triangles ---------------------------------------------------------------------------------------------
    def _export_new(self, handle, tri, tid, tstd):
        """Eager frames (and oversized updates): one small kernel on a side stream writes the three arrays straight into pinned host
        memory; it overlaps the next frame's kernels, which only READ this part of the log."""
        n = tri.size(0)
        self._ensure_pin(n)
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

    def _export_dma(self, sl, tri, n: int):
        """d2h "dma": the frame is complete — its stamp, written behind its snapshot by the last kernel of its extract, has been seen — and its new
        triangles sit in the log (mesh left in HBM).  They go to the frame's pinned slot beside the next frame's kernels, enqueued already:
          * by the SDMA engines (`dif_mesh_cache_export_sdma`: HSA copies on engines the library chooses, three row ranges under one signal, no
            copy kernel on any queue; the call returns when the rows have landed) — unless the process's HSA runtime cannot be reached or is
            not the build the engine selection was validated with (the call says so: DIF_ELAUNCH), or the engines are slow in this process
            (eight small exports in a row above SDMA_SLOW_US; profiles/r05_experiments.md 3);
          * else by three hipMemcpyAsync on a side stream (blit kernels on this runtime).
        (What the copy reads was written by an EARLIER kernel of the frame than the one that stamps: that kernel's end-of-kernel release has
        written its lines back.)  Returns True when the rows are in the slot."""
        lib, cache = _lib.load(), self.map._cache_struct()
        lo = tri.storage_offset() // 9
        if self.sdma is not False and self.force_export is None:
            t0 = time.perf_counter()
            rc = lib.dif_mesh_cache_export_sdma(ctypes.byref(cache), lo, n, sl["out_ptr"][0], sl["out_ptr"][1], sl["out_ptr"][2])
            us = (time.perf_counter() - t0) * 1e6
            self.sdma = (rc == 0)
            self.sdma_us.append((int(n), us))
            if n < (1 << 15):
                self._sdma_slow = self._sdma_slow + 1 if us > self.SDMA_SLOW_US else 0
                if self._sdma_slow >= 8:
                    self.sdma = False
            if rc == 0:
                return True
        with torch.cuda.device(self.device), torch.cuda.stream(self._copy_stream):
            fn = lib.dif_mesh_cache_export if self.force_export == "kernel" else lib.dif_mesh_cache_export_dma
            _lib.check(fn(ctypes.byref(cache), lo, n, sl["out_ptr"][0], sl["out_ptr"][1], sl["out_ptr"][2], _lib.stream_ptr()), "dif_mesh_cache_export_dma")
            sl["export_event"].record()
        sl["export_event"].synchronize()
        return True

    def _finish_frame(self, handle, d2h: str):
        self._export_deferred_now(handle)
        tri, tid, tstd = self.map.extract_mesh_finish(handle)
        if handle.get("deferred"):
            if "export_event" in handle:
                handle["export_event"].synchronize()
            else:                           # carried out by the next frame's kernels, which stamp the slot's `notify` word when the copy is complete
                _lib.spin_until(handle["host_slots"][handle["host_out"]]["notify_np"], 0, handle["stamp"], "deferred triangle export")
        out = (tri, tid, tstd)
        n = tri.size(0)
        if d2h == "new":
            k = handle.get("host_out")
            if k is not None and n <= self.HOST_OUT_TRIANGLES:
                hp = handle["host_slots"][k]["out"]              # already there: written by the frame's kernels / the next frame's first ones
                out = (hp[0][:n], hp[1][:n], hp[2][:n])
            else:
                out = self._export_new(handle, tri, tid, tstd)
        elif d2h == "dma":
            k = handle.get("dma_slot")
            if n and k is not None and n <= self.HOST_OUT_TRIANGLES:
                sl = handle["host_slots"][k]
                self._export_dma(sl, tri, n)
                hp = sl["out"]
                out = (hp[0][:n], hp[1][:n], hp[2][:n])
            elif n:
                out = self._export_new(handle, tri, tid, tstd)
        elif d2h == "full":
            mc = self.map.mesh_cache
            out = (mc.vertices, mc.vertices_flatten_id, mc.vertices_std)
        self.stats.append(dict(self.map.last_counters))
        return out

    def _finish_pending(self, d2h: str):
        """Complete the pending frame; returns its output (None if there was none)."""
        p, self._pending = self._pending, None
        return None if p is None else self._finish_frame(p, d2h)

    @staticmethod
    def _own_storage(o):
        """An output handed back as DEVICE views of the mesh-cache log (d2h "none") gets storage of its own: a log compaction moves the rows the
        views name.  (A frame completed early in the very call that then compacts the log — a map short of room — came back with rows of
        other triangles until round 5.)"""
        return o if (o is None or not o[0].is_cuda) else tuple(x.clone() for x in o)

    # ---- direct frames: the frame's launches are enqueued by two C calls, the host stays one frame ahead of the GPU --------------------------
    # All sizes live in device counters, so a frame's launches have host-independent shapes.  The frame's FIRST kernel reads a 64-byte frame
    # descriptor (two device pointers + pose) from pinned host memory and its LAST kernel writes the counters (and a stamp) into pinned host
    # memory: `dif_integrate_frame` + `dif_extract` enqueue the frame in ~60 us of host time for ~150 us of GPU time, and the host reads frame
    # i-1's results while frame i runs — provided the map has room for the worst-case allocations of the frames in flight (see __init__:
    # otherwise every frame first waits for its predecessor).
    def _extract_rows(self) -> int:
        """Rows of the per-voxel extract buffers (`DenseIndexedMap._extract_rows`: four times the high-water mark of what a frame of this map
        has decoded; an extract that could need more defers itself and the buffers grow here one frame later)."""
        return self.map._extract_rows(self.resolution)

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
        rows = self._extract_rows()
        sig = (m._capacity, m._ws.data_ptr(), m._cache[0].data_ptr(), rows)
        if self._d_sig != sig:                       # (re)build the per-slot buffer descriptors after a re-allocation
            # (re-allocated state may be filled / copied on the caller's stream only: the front-end stream must not run ahead of that)
            self._ov_leave()
            # (a pending deferred export points into the mesh log: carry it out before anything below may re-allocate that log)
            self._export_deferred_now(self._pending)
            self._d_bufs, self._d_tens = [], []
            for sl in self._d_slots:
                tens, buf = m._extract_buffers(self.resolution, self.max_n_triangles, max_vox=rows)
                buf.counters_out = _lib.ptr(sl["counters"])
                self._d_bufs.append(buf)
                self._d_tens.append(tens)
            self._d_sig = (m._capacity, m._ws.data_ptr(), m._cache[0].data_ptr(), rows)
            self._d_w = m.model.packed.weights_struct(dev)
            self._d_lib = _lib.load()
            self._d_args = (self.intr.height, self.intr.width, self.intr.fx, self.intr.fy, self.intr.cx, self.intr.cy)

    def _direct_begin(self, i: int, d2h: str):
        """Everything of a direct frame that precedes its launches: room in the map for the frames in flight, a due log compaction,
        the frame's pinned slot and buffer descriptor, the frame descriptor written.  To be called with the device current.  Returns
        (slot index, slot, extract buffers, export?, output of a frame that had to be completed early or None)."""
        m = self.map
        if m.meshing_thread is not None and m.meshing_thread.is_alive():
            raise RuntimeError("step_direct cannot be mixed with an asynchronous extract_mesh on the same map")
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
            # (a growth re-allocates and copies the map's state on THIS stream: no front end may be running, or start before it has — ADVICE r5)
            self._ov_leave()
            if self._pending is not None:
                out = self._finish_pending(d2h)              # make the bound exact before deciding to grow
        m._ensure_capacity(may_add)
        self._before_frame()
        if m._gc_wanted:
            self._export_deferred_now(self._pending)      # (the pending copy reads log positions that the compaction moves)
            out = self._own_storage(out)
            m._cache_gc()
        self._direct_prepare()
        k = self._d_seq % self.DIRECT_SLOTS
        self._d_seq += 1
        sl = self._d_slots[k]
        export = d2h == "new"
        self._stamp = (self._stamp % 0x3FFFFFFF) + 1
        buf = self._direct_fill(k, export)
        sl["frame_np"][:] = self._d_desc[i]
        return k, sl, buf, export, out

    def _direct_fill(self, k: int, export: bool):
        """The per-frame fields of slot k's extract-buffer descriptor (ONE place: `_direct_begin`, and again by `FusionStreamGroup.step` when
        another map's growth re-allocated this stream's buffers — and with them the descriptors — between begin and launch)."""
        sl, buf = self._d_slots[k], self._d_bufs[k]
        buf.out_tri, buf.out_id, buf.out_std = (_lib.ptr(t) for t in sl["out"]) if export else (None, None, None)
        buf.out_capacity = self.HOST_OUT_TRIANGLES if export else 0
        # Deferred export: this frame's extract leaves the copy of its new triangles (~0.5 MB over PCIe) to the FIRST kernels of the next
        # frame, where it overlaps the point pass instead of lengthening marching cubes.
        # (Only the one-pass marching cubes — resolution <= 4 — and the deferred copy write a frame's triangles BEFORE the kernel that
        # stamps; the two-pass kernels leave them to `k_extract_finish` itself, whose block 0 stamps without waiting for the other
        # workgroups' copies: there the export is always deferred, so that the stamp never stands for rows it does not cover.)
        buf.defer_export = 1 if (export and (self.defer_export or self.resolution > 4)) else 0
        # Completion without events: the extract's last kernel stamps the pinned counter snapshot, and the kernels that carry the deferred
        # copy out stamp `notify` — the host polls those words instead of waiting on events (an event record costs the queue ~5 us
        # between two kernels, twice per frame).
        buf.stamp = self._stamp
        buf.export_notify = _lib.ptr(sl["notify"]) if buf.defer_export else None
        return buf

    def _direct_integrated(self):
        """Right behind the frame's integrate launches: the PREVIOUS frame's deferred triangle export has just been carried out by this
        frame's point kernels (its fusion kernel stamps the slot's `notify` word)."""
        p = self._pending
        if isinstance(p, dict) and p.get("deferred") and "export_event" not in p:
            p["export_carried"] = True

    def _direct_end(self, k, sl, buf, export, d2h, out):
        """Behind the frame's extract launches: the frame's handle; completes the previous frame on the host."""
        m = self.map
        m.mesh_cache.invalidate_host_copy()
        h = dict(stamp=int(buf.stamp), counters=sl["counters_np"], epoch=m._gc_epoch, add_total=m._add_total, max_n_triangles=self.max_n_triangles,
                 host_out=(k if export else None), host_slots=self._d_slots, deferred=bool(buf.defer_export))
        if d2h == "dma":
            h["dma_slot"] = k
        done = self._finish_pending(d2h)
        if done is not None:
            out = done
        self._pending = h
        return out

    def step_direct(self, i: int, d2h: str = "new"):
        """One frame enqueued with two C calls, host one frame ahead; returns the previous frame's output like `step_pipelined`."""
        m = self.map
        with torch.cuda.device(self.device):
            sp = _lib.stream_ptr()
            # two queues only where `enable_overlap` found them independent (the caller's stream of THEN), for the exports that need no kernel of
            # the next frame, and for the extract configuration whose every kernel is covered by the two-queue tests
            ov = bool(self.overlap and d2h in ("dma", "none") and self.resolution <= 4 and sp.value == self._ov_main)
            if not ov:
                self._ov_leave()
            k, sl, buf, export, out = self._direct_begin(i, d2h)
            lib, w = self._d_lib, self._d_w
            H, W, fx, fy, cx, cy = self._d_args
            self.last_tensors = self._d_tens[k]
            cm = m._cmap
            if ov:
                # this frame's front end on `_fe_stream` (behind the previous frame's fusion kernel, beside its extract), its fusion kernel behind
                # that extract on the caller's stream, its extract behind its fusion kernel: dif_map_t.frame_seq
                if not self._ov_active:
                    self._ov_enter()
                self._ov_seq += 1
                cm.frame_seq, cm.fuse_stream, cm.frame_counters = self._ov_seq, sp, _lib.ptr(m._frame_counters)
            try:
                _lib.check(lib.dif_integrate_frame(ctypes.byref(cm), ctypes.byref(w), _lib.ptr(sl["frame"]), H, W, fx, fy, cx, cy, *self._pts(),
                                                   _lib.ptr(self._d_mask), _lib.ptr(m._ws), m._ws.numel(), self._fe_ptr if ov else sp), "dif_integrate_frame")
                self._direct_integrated()
                if self.tiling is not None:
                    self._exchange_halo(reserved=True)              # export -> send/recv -> merge, all on this stream, no host wait
                _lib.check(lib.dif_extract(ctypes.byref(cm), ctypes.byref(w), ctypes.byref(buf), int(self.resolution), 1, float(self.max_std), 0, 1, sp),
                           "dif_extract")
            finally:
                cm.frame_seq, cm.fuse_stream, cm.frame_counters = 0, None, None
            return self._direct_end(k, sl, buf, export, d2h, out)


class FusionStreamGroup:
    """S independent subsequences on ONE GPU whose frames share their launches (`dif_integrate_frames` + `dif_extract_streams`):
    BASELINE config C4's unit of work — an independent stream with a private map — batched inside a GPU.  A steady-state frame of one
    stream is 1-2 MLP tiles per SIMD and latency-bound launches; S of them per launch fill the SIMDs and pay each floor once.  Every stream
    keeps its own map, pinned slots, deferred export and host bookkeeping (`FusionStream`), and its results are bit-identical to stepping
    it alone (`tests/test_gpu_stream.py::test_stream_group_matches_single_streams`)."""

    MIN_CAPACITY = 8192         # dif_extract_streams' dirty-set scan walks whole 256-slot blocks of maps with more than 4,096 slots

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

    def _equalise(self):
        """The batched launches are shaped by ONE capacity and ONE number of extract rows: a map that has grown (or wants more rows) pulls
        the others along (rare: the capacity covers three frames of worst-case allocations, see FusionStream.__init__)."""
        cap = max(self.MIN_CAPACITY, max(st.map._capacity for st in self.streams))
        hw = max(st.map._extract_high_water for st in self.streams)
        want = max(st.map._extract_rows_wanted for st in self.streams)
        changed = False
        for st in self.streams:
            if st.map._capacity != cap:
                st._export_deferred_now(st._pending)
                with st.map._state_lock:
                    st.map._alloc_state(cap)
                changed = True
            if st.map._extract_high_water != hw or st.map._extract_rows_wanted != want:
                st.map._extract_high_water, st.map._extract_rows_wanted = hw, want
                changed = True
        return changed

    def step(self, idx, d2h: str = "new"):
        """Frame idx[j] (or the same index for all, if an int) of every stream j, enqueued with two C calls for the whole group; returns the
        list of the streams' previous-frame outputs (None entries on the first call), like `FusionStream.step_direct`."""
        S = len(self.streams)
        idx = [idx] * S if isinstance(idx, int) else list(idx)
        a = self.streams[0]
        with torch.cuda.device(self.device):
            for st in self.streams:
                st._ov_leave()                      # (S streams per launch fill the machine as they are: a group's frames stay on one queue)
            self._equalise()
            begun = [st._direct_begin(i, d2h) for st, i in zip(self.streams, idx)]
            if self._equalise():                    # a map grew (or a frame completed early reported a deferral) inside _direct_begin
                for st in self.streams:
                    st._direct_prepare()            # descriptors of re-allocated buffers
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
                st.last_tensors = st._d_tens[k]
            lib, w, sp = a._d_lib, a._d_w, _lib.stream_ptr()
            H, W, fx, fy, cx, cy = a._d_args
            _lib.check(lib.dif_integrate_frames(self._frames, S, ctypes.byref(w), H, W, fx, fy, cx, cy, sp), "dif_integrate_frames")
            for st in self.streams:
                st._direct_integrated()
            _lib.check(lib.dif_extract_streams(self._frames, S, ctypes.byref(w), int(a.resolution), float(a.max_std), 1, sp), "dif_extract_streams")
            return [st._direct_end(k, sl, buf, export, d2h, out) for st, (k, sl, buf, export, out) in zip(self.streams, begun)]

    def flush(self, d2h: str = "new"):
        return [st.flush(d2h) for st in self.streams]
