"""`DenseIndexedMap` — the map / integrate / extract surface of the reference's `pytorch/system/map.py:158-723`,
backed by libdifusion.so (hand-written HIP for gfx950).  Same constructor, attributes, methods and return values;
the tensors behind the properties live on the GPU and are owned here, the kernels receive raw pointers.

Differences a caller can observe (all documented in DESIGN.md):
  * `integrate_keyframe` makes no host round trip (the reference syncs 4 times, `map.py:382,441-444`) as long as the host-side
    upper bound of `n_occupied` stays below the capacity; the bound is made exact by every finished extract (async pinned copy of
    the counters) and, failing that, by one blocking counter read before the buffers are doubled;
  * `n_occupied` is a device counter — reading the property synchronises;
  * `latent_vecs`, `latent_vecs_pos`, `voxel_obs_count`, `voxel_optimized` are views of larger pre-allocated buffers,
    sliced to the capacity the reference's doubling rule (`map.py:263-285`) would have reached;
  * points outside the map bounds / NaN points are ignored instead of indexing out of range (`map.py:313`);
  * triangles come out in a canonical order (voxel, cell, table) instead of atomic arrival order;
  * `do_optimize=True` runs the latent optimisation synchronously on the GPU (`async_optimize=True`, the visualisers and
    `interpolate=False` raise NotImplementedError: out of scope / dead in the reference — `system.ext.marching_cubes` does not exist
    there either, `map.py:693`).
"""
from __future__ import annotations

import argparse
import ctypes
import functools
import logging
import threading
from pathlib import Path
from typing import Optional

import numpy as np
import torch

from .. import _lib
from ..network import utility as net_util


class MeshExtractCache:
    """reference `map.py:116-133`.  The three arrays live in HBM (two ping-pong buffer sets maintained by `dif_extract`,
    same content and ORDER as the reference's host arrays); the attributes copy them to the host on demand.
    `updated_vec_id` is a dirty flag per slot on the GPU (`DenseIndexedMap.updated_vec_id`)."""

    def __init__(self, device, owner=None):
        self.device = device
        self._owner = owner
        self._host = None

    def _fetch(self):
        if self._host is None:
            t = self._owner.mesh_cache_tensors()
            self._host = None if t is None else tuple(x.cpu().numpy() for x in t)
        return self._host

    @property
    def vertices(self):
        h = self._fetch()
        return None if h is None else h[0]

    @property
    def vertices_flatten_id(self):
        h = self._fetch()
        return None if h is None else h[1]

    @property
    def vertices_std(self):
        h = self._fetch()
        return None if h is None else h[2]

    def invalidate_host_copy(self):
        self._host = None

    def clear_all(self):
        self._host = None
        self._owner._cache_clear()


class Mesh:
    """Minimal stand-in for `open3d.geometry.TriangleMesh` (Open3D is optional): what `_make_mesh_from_cache`
    (`map.py:521-543`) would put into it."""

    def __init__(self, vertices: np.ndarray, triangles: np.ndarray, vertex_std: np.ndarray):
        self.vertices = vertices            # (3T, 3) float64
        self.triangles = triangles          # (T, 3) int32
        self.vertex_std = vertex_std        # (3T,) float


class _GetSdfFn(torch.autograd.Function):
    """sdf(xyz) with the kernel-computed Jacobian d sdf_m / d xyz_{sel[m]} (each output depends on one input point only)."""

    @staticmethod
    def forward(ctx, xyz, owner):
        sdf, std, mask, sel, grad = owner._query(xyz, True, keep_inverse=True)
        inv = owner._query_inverse                          # this query's own inverse map (point -> row among the valid ones)
        ctx.save_for_backward(sel, grad, inv)
        ctx.n = xyz.size(0)
        ctx.mark_non_differentiable(std, mask)
        return sdf, std, mask

    @staticmethod
    def backward(ctx, g_sdf, g_std, g_mask):
        sel, grad, inv = ctx.saved_tensors
        n = ctx.n
        if n == 0 or grad.size(0) == 0:
            return torch.zeros((n, 3), dtype=torch.float32, device=grad.device), None
        out = torch.empty((n, 3), dtype=torch.float32, device=grad.device)
        g = g_sdf.contiguous().float()
        with _on_device(grad.device):                 # ONE launch, every row written: out[i] = grad[inv[i]] * g_sdf[inv[i]], or 0 for an invalid point
            _lib.check(_lib.load().dif_query_grad_gather(_lib.ptr(grad), _lib.ptr(g), _lib.ptr(inv), n, _lib.ptr(out), _lib.stream_ptr()),
                       "dif_query_grad_gather")
        return out, None


_OVERFLOW_WHAT = {8: "a delta halo message overflowed: the neighbours' halo copies are stale from that frame on (use whole-layer messages, "
                     "parallel.exchange_halo(mode='full'), for streams that change this much per frame)",
                  1: "more voxels than latent rows", 2: "more dirty voxels than extract buffers", 3: "more decoded voxels than extract buffers",
                  5: "mesh-cache log full", 6: "more records than the export buffer",
                  7: "marching cubes gave up waiting for an earlier workgroup (the GPU was shared with another kernel for seconds)"}


class _NoSwitch:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NO_SWITCH = _NoSwitch()


def _on_device(dev):
    """`torch.cuda.device(dev)` only when `dev` is not the current device already (the context manager costs ~10 us of host time per use)."""
    if torch.cuda.current_device() == (dev.index if dev.index is not None else 0):
        return _NO_SWITCH
    return torch.cuda.device(dev)


def _next_pow2(n: int) -> int:
    p = 1
    while p < n:
        p *= 2
    return p


class DenseIndexedMap:
    def __init__(self, model: net_util.Networks, args: argparse.Namespace, latent_dim: int, device: torch.device,
                 enable_async: bool = False, optimization_device: torch.device = None, initial_capacity: int = 1 << 16):
        """reference `map.py:159-234`.  `initial_capacity`: rows pre-allocated for latent vectors (grows by doubling)."""
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("DenseIndexedMap runs on the GPU only (libdifusion has no CPU fallback)")
        if latent_dim != _lib.LATENT_DIM:
            raise NotImplementedError("libdifusion is specialised to latent_dim = 29 (ckpt/default/hyper.json)")
        self.model = model
        self.model.eval()
        self.voxel_size = args.voxel_size
        self.n_xyz = np.ceil((np.asarray(args.bound_max) - np.asarray(args.bound_min)) / args.voxel_size).astype(int).tolist()
        logging.info(f"Map size Nx = {self.n_xyz[0]}, Ny = {self.n_xyz[1]}, Nz = {self.n_xyz[2]}")
        self.args = args
        self.bound_min = torch.tensor(args.bound_min, device=device).float()
        self.bound_max = self.bound_min + self.voxel_size * torch.tensor(self.n_xyz, device=device)
        self.latent_dim = latent_dim
        self.device = device
        self.extract_mesh_std_range = None
        self.modifying_lock = threading.Lock()
        # host-side bookkeeping (allocation bound, counter snapshots) is touched by the meshing thread and by the integrating thread
        self._state_lock = threading.RLock()
        self._integrate_done = None         # event recorded behind the last integrate on ITS stream: extracts wait for it
        self.meshing_thread = None
        self.meshing_thread_id = -1
        self.meshing_stream = torch.cuda.Stream(device=device)
        self.mesh_cache = MeshExtractCache(self.device, self)
        self._cache = None                  # mesh-cache log: (tri, id, std, alive) device buffers
        self._cache_out = None              # compaction target (and the next log after a garbage collection)
        self._cache_cur = 0                 # kept for API compatibility of handles; the log is a single buffer
        self._cache_any = False             # an extract has produced a cache
        self._gc_epoch = 0
        self._gc_log_len = 0
        self._gc_wanted = False
        self._halo_scratch = None
        self._opt_ws = None
        self.optimize_noise = None          # optional (k,) tensor of N(0,1) samples for the optimiser's perturbations (default: torch.randn)
        self.optimize_losses = None         # device float[64]: likelihood loss before each Adam step of the last optimisation
        self._cache_call_limit = 0          # max_n_triangles of the latest extract (what one more call may append to the log)
        self.extract_buffer_bytes = 8 << 30 # upper bound for the per-voxel extract buffers (see _extract_rows)
        self._extract_high_water = 0        # most voxels any extract of this map has decoded or meshed (max over frames of B, K)
        self._extract_rows_wanted = 0       # rows the last DEFERRED extract would have needed (counters[DIF_C_DEFERRED]); 0 = none deferred
        self.n_deferred = 0                 # extracts that deferred themselves (their dirty set was meshed by a later one)
        self._query_ws = None               # get_sdf: scan scratch + pinned count slots
        self._halo_list = None              # spatial tiling: boundary change lists (dif_map_t.halo_list)
        self._halo_lists_stale = True       # the lists do not cover every change since the last halo export: whole-layer messages next

        self._grid = int(np.prod(self.n_xyz))
        if self._grid >= 2 ** 31:
            raise RuntimeError("grid too large for 32-bit linear ids")
        with torch.cuda.device(device):
            self._indexer = torch.full((self._grid,), -1, device=device, dtype=torch.long)
            self._frame_count = torch.zeros((self._grid,), device=device, dtype=torch.int32)
            self._grid_bits = torch.zeros(((self._grid + 31) // 32,), device=device, dtype=torch.int32)
            self._grid_tot = torch.zeros((1024,), device=device, dtype=torch.int32)
            # the allocation scan's own bitmap (dif_map_t.alloc_bits): the extract's neighbourhood marker keeps grid_bits, so a frame's integrate
            # front end may run beside the previous frame's extract (two queues, FusionStream.overlap)
            self._alloc_bits = torch.zeros(((self._grid + 31) // 32,), device=device, dtype=torch.int32)
            self._alloc_tot = torch.zeros((1024,), device=device, dtype=torch.int32)
            self._sync_words = torch.zeros((_lib.SYNC_WORDS,), device=device, dtype=torch.int32)     # dif_map_t.sync_words
            self._frame_counters = torch.zeros((_lib.FC_COUNT,), device=device, dtype=torch.int32)    # dif_map_t.frame_counters
            self._counters = torch.zeros((_lib.C_COUNT,), device=device, dtype=torch.int32)
            self._pending_export = torch.zeros((32,), device=device, dtype=torch.int32)        # dif_pending_export_t (72 bytes), idle all-zero
        self._capacity = 0
        self._alloc_state(_next_pow2(max(int(initial_capacity), 1024)))
        self._n_occ_ub = 0                  # host-side upper bound of n_occupied (exact after a counter read)
        self._ws = None
        self._ws_n = 0
        self._xbuf = None
        self._host_counters = (ctypes.c_int32 * _lib.C_COUNT)()
        self._pinned_counters = None
        self._pending_seq = 0
        self._add_total = 0                 # running sum of the per-call allocation bounds (see _ensure_capacity)
        self.last_counters = {}

    # ---- state ------------------------------------------------------------------------------------------------
    def _alloc_state(self, capacity: int):
        dev = self.device
        with torch.cuda.device(dev):
            lat = torch.zeros((capacity, self.latent_dim), dtype=torch.float32, device=dev)
            pos = torch.full((capacity,), -1, dtype=torch.long, device=dev)
            obs = torch.zeros((capacity,), dtype=torch.float32, device=dev)
            dirty = torch.zeros((capacity,), dtype=torch.uint8, device=dev)
            optimized = torch.zeros((capacity,), dtype=torch.uint8, device=dev)
            vbm = torch.full((capacity,), -1, dtype=torch.int32, device=dev)
            rec_dir = torch.zeros((capacity, 16), dtype=torch.int32, device=dev)
            upd_list = torch.zeros((capacity,), dtype=torch.int32, device=dev)
            tri_start = torch.zeros((capacity,), dtype=torch.int32, device=dev)
            tri_n = torch.zeros((capacity,), dtype=torch.int32, device=dev)
            self._dirty_tot = torch.zeros(((capacity + 255) // 256,), dtype=torch.int32, device=dev)       # set dirty flags per 256 slots
            if self._capacity > 0:
                c = self._capacity
                lat[:c] = self._latent
                pos[:c] = self._pos
                obs[:c] = self._obs
                dirty[:c] = self._dirty
                optimized[:c] = self._optimized
                tri_start[:c] = self._tri_start
                tri_n[:c] = self._tri_n
        self._latent, self._pos, self._obs, self._dirty, self._optimized = lat, pos, obs, dirty, optimized
        self._tri_start, self._tri_n = tri_start, tri_n
        self._vbm, self._rec_dir, self._upd_list = vbm, rec_dir, upd_list
        self._capacity = capacity
        m = _lib.DifMap()
        m.nx, m.ny, m.nz = self.n_xyz
        bm = [float(np.float32(v)) for v in self.args.bound_min]
        m.bound_min = (ctypes.c_float * 3)(*bm)
        m.voxel_size = float(self.voxel_size)
        m.prune_min_vox_obs = int(self.args.prune_min_vox_obs)
        m.ignore_count_th = float(self.args.ignore_count_th)
        m.encoder_count_th = float(self.args.encoder_count_th)
        m.capacity = capacity
        m.indexer = _lib.ptr(self._indexer)
        m.latent_vecs = _lib.ptr(lat)
        m.latent_vecs_pos = _lib.ptr(pos)
        m.voxel_obs_count = _lib.ptr(obs)
        m.dirty = _lib.ptr(dirty)
        m.voxel_optimized = _lib.ptr(optimized)
        m.counters = _lib.ptr(self._counters)
        m.frame_count = _lib.ptr(self._frame_count)
        m.grid_bits = _lib.ptr(self._grid_bits)
        m.grid_tot = _lib.ptr(self._grid_tot)
        m.vbm = _lib.ptr(vbm)
        m.rec_dir = _lib.ptr(rec_dir)
        m.upd_list = _lib.ptr(upd_list)
        m.tri_start = _lib.ptr(tri_start)
        m.tri_n = _lib.ptr(tri_n)
        m.own_x_lo, m.own_x_hi, m.halo = getattr(self, "_ownership", (0, self.n_xyz[0], 0))
        m.dirty_tot = _lib.ptr(self._dirty_tot)
        m.pending_export = _lib.ptr(self._pending_export)
        hl = getattr(self, "_halo_list", None)
        m.halo_list = _lib.ptr(hl)
        m.halo_list_cap = 0 if hl is None else hl.size(1)
        m.alloc_bits = _lib.ptr(self._alloc_bits)
        m.alloc_tot = _lib.ptr(self._alloc_tot)
        m.sync_words = _lib.ptr(self._sync_words)
        m.frame_seq = 0                     # two queues off; FusionStream sets it (and fuse_stream) per overlapped frame
        m.fuse_stream = None
        m.frame_counters = None
        self._cmap = m
        self._recount_dirty()

    def _recount_dirty(self):
        """`dif_map_t.dirty_tot` from the flags themselves: after everything that sets dirty flags other than an integrate (which keeps
        the totals itself), and after a re-allocation."""
        cap = self._capacity
        with torch.cuda.device(self.device):
            n_blk = (cap + 255) // 256
            flags = self._dirty
            if n_blk * 256 != cap:
                flags = torch.nn.functional.pad(flags, (0, n_blk * 256 - cap))
            self._dirty_tot.copy_(flags.view(n_blk, 256).sum(dim=1, dtype=torch.int32))

    def _publish_counters(self, c, add_total_at_read):
        with self._state_lock:
            return self._publish_counters_locked(c, add_total_at_read)

    def _publish_counters_locked(self, c, add_total_at_read, clear_flag=False):
        if c[_lib.C_OVERFLOW] != 0:
            # Reported once, so that the session can go on (e.g. with a no_cache re-extraction).  A snapshot taken by an extract has
            # already cleared the device flag in stream order, right behind the copy (`k_extract_finish` / `extract_mesh_enqueue`): the
            # next frame's snapshot neither repeats this overflow nor loses one raised in between.  Only a blocking read clears it here.
            if clear_flag:
                with torch.cuda.device(self.device):
                    self._counters[_lib.C_OVERFLOW:_lib.C_OVERFLOW + 1].zero_()
            raise RuntimeError(f"libdifusion: device buffer overflow (code {c[_lib.C_OVERFLOW]}: "
                               f"{_OVERFLOW_WHAT.get(c[_lib.C_OVERFLOW], '?')}); the result of that call is incomplete")
        # exact n_occupied at the time the counters were read + whatever later calls may have allocated since
        self._n_occ_ub = c[_lib.C_N_OCCUPIED] + (self._add_total - add_total_at_read)
        self.last_counters = dict(n_occupied=c[_lib.C_N_OCCUPIED], alloc_new=c[_lib.C_ALLOC_NEW], M=c[_lib.C_M], C=c[_lib.C_C],
                                  items=c[_lib.C_ITEMS], K=c[_lib.C_K], B=c[_lib.C_B], VH=c[_lib.C_VH], T=c[_lib.C_T],
                                  query_M=c[_lib.C_QUERY_M], opt_rows=c[_lib.C_OPT_ROWS], opt_voxels=c[_lib.C_OPT_VOXELS],
                                  cache_T=c[_lib.C_CACHE_T], cache_kept=c[_lib.C_CACHE_KEPT],
                                  cache_dead=c[_lib.C_CACHE_DEAD], cache_live=c[_lib.C_CACHE_LIVE], deferred=c[_lib.C_DEFERRED])
        self._extract_high_water = max(self._extract_high_water, c[_lib.C_B], c[_lib.C_K])
        return self.last_counters

    def _read_counters(self):
        """Blocking read of the device counters on the current stream.  Every integrate is enqueued completely while `_state_lock` is
        held, so with the lock taken and — when a second thread / stream is in play — the whole device drained, no allocation that
        `_add_total` already accounts for can still be missing from the value read here."""
        with self._state_lock, torch.cuda.device(self.device):
            if self.meshing_thread is not None or threading.current_thread() is not threading.main_thread():
                torch.cuda.synchronize(self.device)
            _lib.check(_lib.load().dif_read_counters(ctypes.byref(self._cmap), self._host_counters, _lib.stream_ptr()), "dif_read_counters")
            return self._publish_counters_locked(list(self._host_counters), self._add_total, clear_flag=True)

    def _ensure_capacity(self, may_add: int):
        """Called with `_state_lock` held by whoever is about to enqueue work that may allocate up to `may_add` voxels."""
        with self._state_lock:
            if self._n_occ_ub + may_add > self._capacity:
                self._read_counters()                              # make the bound exact
                if self._n_occ_ub + may_add > self._capacity:
                    self._alloc_state(_next_pow2(self._n_occ_ub + may_add))
            self._n_occ_ub += may_add
            self._add_total += may_add

    def _ref_capacity(self) -> int:
        """Buffer length the reference would have after the same allocations (doubling from 1, map.py:263-268)."""
        return max(1, _next_pow2(self.n_occupied))

    # properties of map.py:199-220
    @property
    def n_occupied(self) -> int:
        return int(self._read_counters()["n_occupied"])

    @property
    def indexer(self) -> torch.Tensor:
        return self._indexer

    @property
    def latent_vecs(self) -> torch.Tensor:
        return self._latent[:self._ref_capacity()]

    @property
    def latent_vecs_pos(self) -> torch.Tensor:
        return self._pos[:self._ref_capacity()]

    @property
    def voxel_obs_count(self) -> torch.Tensor:
        return self._obs[:self._ref_capacity()]

    @property
    def voxel_optimized(self) -> torch.Tensor:
        return self._optimized[:self._ref_capacity()].view(torch.bool)

    @property
    def updated_vec_id(self) -> torch.Tensor:
        """`mesh_cache.updated_vec_id` of the reference (sorted slot ids awaiting re-meshing, map.py:303-308)."""
        n = self.n_occupied
        return torch.nonzero(self._dirty[:n]).flatten()

    @property
    def cold_vars(self):
        n = self.n_occupied
        c = self._ref_capacity()
        return {"n_occupied": n, "indexer": self._indexer, "latent_vecs": self._latent[:c], "latent_vecs_pos": self._pos[:c],
                "voxel_obs_count": self._obs[:c], "voxel_optimized": self._optimized[:c].view(torch.bool)}

    def save(self, path):
        """reference `map.py:239-243`: `torch.save` of the cold_vars dict (load-compatible with the reference)."""
        path = Path(path)
        with path.open("wb") as f:
            torch.save({k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in self.cold_vars.items()}, f)

    def load(self, path):
        """reference `map.py:245-249`; accepts maps saved by the reference or by `save`."""
        path = Path(path)
        with path.open("rb") as f:
            cv = torch.load(f, map_location=self.device)
        n = int(cv["n_occupied"])
        if cv["indexer"].numel() != self._grid:
            raise RuntimeError("saved map has a different grid")
        if n > self._capacity:
            self._alloc_state(_next_pow2(n))
        self._indexer.copy_(cv["indexer"].view(-1))
        self._latent.zero_(); self._pos.fill_(-1); self._obs.zero_(); self._dirty.zero_(); self._optimized.zero_()
        if "voxel_optimized" in cv:
            self._optimized[:n] = cv["voxel_optimized"][:n].to(torch.uint8)
        self._latent[:n] = cv["latent_vecs"][:n]
        self._pos[:n] = cv["latent_vecs_pos"][:n]
        self._obs[:n] = cv["voxel_obs_count"][:n]
        self._counters.zero_()
        self._counters[_lib.C_N_OCCUPIED] = n
        self._n_occ_ub = n
        self._halo_lists_stale = True
        self._recount_dirty()
        self.mesh_cache.clear_all()

    # ---- integrate --------------------------------------------------------------------------------------------
    def integrate_keyframe(self, surface_xyz: torch.Tensor, surface_normal: torch.Tensor, do_optimize: bool = False,
                           async_optimize: bool = False):
        """reference `map.py:340-519`.  (N,3) xyz + (N,3) normals, float32 on the map's device.
        :return: unq_mask (N,) bool — points whose voxel holds more than `prune_min_vox_obs` points (None if pruning is off)."""
        assert surface_xyz.device == surface_normal.device == self.device, \
            f"Device of map {self.device} and input observation {surface_xyz.device, surface_normal.device} must be the same."
        if do_optimize and async_optimize:
            raise NotImplementedError("the latent optimisation runs synchronously here (on the GPU, a few kernel launches); the reference's "
                                      "separate optimiser process (map.py:28-78, 507-510) is not reproduced")
        xyz = surface_xyz.contiguous().float()
        nrm = surface_normal.contiguous().float()
        N = xyz.size(0)
        lib = _lib.load()
        with self.modifying_lock, self._state_lock, torch.cuda.device(self.device):
            torch.cuda.current_stream().wait_stream(self.meshing_stream)
            prune = int(self.args.prune_min_vox_obs)
            self._ensure_capacity(7 * (N // (prune + 1)) if prune > 0 else 7 * N)
            if self._ws is None or self._ws_n < N:
                nb = int(lib.dif_integrate_workspace_bytes(N))
                if nb < 0:
                    raise RuntimeError("dif_integrate_workspace_bytes failed")
                self._ws = torch.empty((nb,), dtype=torch.uint8, device=self.device)
                self._ws_n = N
            mask = torch.empty((N,), dtype=torch.uint8, device=self.device)
            w = self.model.packed.weights_struct(self.device)
            _lib.check(lib.dif_integrate(ctypes.byref(self._cmap), ctypes.byref(w), _lib.ptr(xyz), _lib.ptr(nrm), N, _lib.ptr(mask),
                                         _lib.ptr(self._ws), self._ws.numel(), _lib.stream_ptr()), "dif_integrate")
            if do_optimize and int(getattr(self.args, "optim_n_iters", 0)) > 0:
                self._optimize_latents(lib, w, xyz, nrm, N, mask)
            if self._integrate_done is None:
                self._integrate_done = torch.cuda.Event()
            self._integrate_done.record()                          # on the integrating stream, whichever it is
        return mask.view(torch.bool) if int(self.args.prune_min_vox_obs) > 0 else None

    def _optimize_latents(self, lib, w, xyz, nrm, N, mask):
        """Stage 3 of `integrate_keyframe(do_optimize=True)` (reference `map.py:459-513`, `OptimizeProcess.do_optimize` `:80-113`,
        `_update_optimize_result_set(deintegrate_old=False)` `:321-335`): `dif_optimize_latents`.  The perturbation samples come from
        `torch.randn` on the map's device (the reference draws them the same way); `optimize_noise`, if set, is used instead (tests feed
        the numbers the reference drew)."""
        nb = int(lib.dif_optimize_workspace_bytes(N, self._capacity))
        if self._opt_ws is None or self._opt_ws.numel() < nb:
            self._opt_ws = torch.empty((nb,), dtype=torch.uint8, device=self.device)
        noise = self.optimize_noise
        if noise is None:
            noise = torch.randn((8 * N,), dtype=torch.float32, device=self.device)
        else:
            noise = torch.cat([noise.to(self.device).float().flatten(), torch.zeros((8 * N,), device=self.device)])[:8 * N].contiguous()
        self.optimize_losses = torch.zeros((64,), dtype=torch.float32, device=self.device)
        lam = float(self.args.code_reg_lambda) if getattr(self.args, "code_regularization", False) else 0.0
        _lib.check(lib.dif_optimize_latents(ctypes.byref(self._cmap), ctypes.byref(w), _lib.ptr(xyz), _lib.ptr(nrm), N, _lib.ptr(mask), _lib.ptr(noise),
                                            int(self.args.optim_n_iters), 1.0e-2, lam, _lib.ptr(self.optimize_losses), _lib.ptr(self._opt_ws),
                                            self._opt_ws.numel(), _lib.stream_ptr()), "dif_optimize_latents")
        self._halo_lists_stale = True                            # changed voxels that are not in the boundary change lists
        self._recount_dirty()                                    # the write-back marks the optimised voxels dirty

    def allocate_block(self, idx: torch.Tensor):
        """reference `map.py:310-319`.  Slots are handed out in ASCENDING linear-id order (the only order the reference's
        own caller ever passes, `map.py:383-387`); ids that are already allocated are left untouched (weight-0 records)."""
        if idx.ndimension() == 2 and idx.size(1) == 3:
            idx = idx[:, 2] + self.n_xyz[-1] * idx[:, 1] + (self.n_xyz[-1] * self.n_xyz[-2]) * idx[:, 0]
        # slots are numbered in ascending id order, whatever order the caller passes (the reference numbers them in the caller's order,
        # map.py:317-319, and its only caller passes torch.unique output, i.e. sorted unique ids: same numbering there)
        idx = torch.unique(idx.to(self.device).long().flatten())
        if idx.numel() == 0:
            return
        rec = torch.zeros((idx.size(0), 32), dtype=torch.int32, device=self.device)
        rec[:, 0] = idx.to(torch.int32)          # grid < 2^31 (checked in __init__), high word stays 0
        self.merge_records(rec)

    # ---- get_sdf ----------------------------------------------------------------------------------------------
    def _query(self, xyz: torch.Tensor, want_grad: bool, keep_inverse: bool = False):
        """`dif_query_select` (mask + ordered compaction of the valid points), then `dif_query_decode` over the M selected rows.  The host
        needs M to hand back M-row tensors (the reference's return shapes): the compaction's last workgroup writes it into pinned host
        memory and the host waits for THAT small kernel only — the decoder is still running when this returns (the tracker calls
        get_sdf once per Gauss-Newton iteration, reference tracker.py:184: no GPU idle gap between iterations)."""
        xyz = xyz.detach().contiguous().float()
        _lib.require_cuda(xyz)
        N = xyz.size(0)
        dev = self.device
        lib = _lib.load()
        # (the tracker calls this once per Gauss-Newton iteration and waits for the result: what counts here is HOST time per call — one
        # allocation for the five outputs, no device-context switch when the map's device is current already, no event in the queue)
        with _on_device(dev):
            n1 = max(N, 1)
            blob = torch.empty((n1 * (6 if want_grad else 3) + (n1 + 3) // 4,), dtype=torch.float32, device=dev)
            sdf, std = blob[:n1], blob[n1:2 * n1]
            sel = blob[2 * n1:3 * n1].view(torch.int32)
            grad = blob[3 * n1:6 * n1].view(n1, 3) if want_grad else None
            mask = blob[(6 if want_grad else 3) * n1:].view(torch.uint8)[:N]
            if keep_inverse:
                # (the autograd path keeps the compaction's inverse map for its backward: a scratch of its own per call, so that a later query
                # cannot overwrite it; the pinned count slots stay shared)
                self._query_inverse = torch.empty((N + 4096,), dtype=torch.int32, device=dev)
            if N == 0:
                return sdf[:0], std[:0], mask.view(torch.bool), sel[:0], (grad[:0] if want_grad else None)
            q = self._query_ws
            if q is None or q["scratch"].numel() < N + 4096:
                # (per-map, grow-only: scan scratch and the pinned slots the count comes back through)
                q = self._query_ws = dict(scratch=torch.empty((N + 4096,), dtype=torch.int32, device=dev),
                                          notes=[torch.zeros((2,), dtype=torch.int32).pin_memory() for _ in range(4)],
                                          events=[torch.cuda.Event() for _ in range(4)], seq=0)
                for t in q["notes"]:
                    t.numpy()[1] = -1
                q["notes_np"] = [t.numpy() for t in q["notes"]]
            q["seq"] = seq = (q["seq"] + 1) & 0x3FFFFFFF
            k = seq & 3
            note, ev = q["notes"][k], q["events"][k]
            sp = _lib.stream_ptr()
            _lib.check(lib.dif_query_select(ctypes.byref(self._cmap), _lib.ptr(xyz), N, _lib.ptr(mask), _lib.ptr(sel),
                                            _lib.ptr(self._query_inverse if keep_inverse else q["scratch"]), _lib.ptr(note), seq, sp), "dif_query_select")
            w = self.model.packed.weights_struct(dev)
            _lib.check(lib.dif_query_decode(ctypes.byref(self._cmap), ctypes.byref(w), _lib.ptr(xyz), N, _lib.ptr(sel), _lib.ptr(sdf), _lib.ptr(std),
                                            _lib.ptr(grad), sp), "dif_query_decode")
            # the compaction's last workgroup writes (M, seq) into the pinned note — M first, then a system-scope fence, then seq — and the host
            # polls for ITS sequence number: it waits for the compaction only (the decoder keeps running), with no event in the queue
            got = q["notes_np"][k]
            _lib.spin_until(got, 1, seq, "get_sdf's compaction")
            M = int(got[0])
        return sdf[:M], std[:M], mask.view(torch.bool), sel[:M], (grad[:M] if want_grad else None)

    def get_sdf(self, xyz: torch.Tensor):
        """reference `map.py:559-579`: (N,3) -> sdf (M,), std (M,), valid_mask (N,) bool.
        If `xyz.requires_grad`, `sdf` carries an autograd edge back to `xyz` (analytic d sdf / d xyz from the decoder's reverse
        MFMA chain), which is what `SDFTracker.compute_sdf_Hg` differentiates (reference `tracker.py:186-192`: the loss is
        sdf / std.detach(), so `std` is returned without a graph)."""
        if torch.is_grad_enabled() and xyz.requires_grad:
            sdf, std, mask = _GetSdfFn.apply(xyz, self)
            return sdf, std, mask
        sdf, std, mask, _, _ = self._query(xyz, False)
        return sdf, std, mask

    def get_sdf_with_gradient(self, xyz: torch.Tensor):
        """`get_sdf` plus the analytic d sdf / d xyz of the valid points, (M, 3), straight from the decoder's reverse chain — for callers
        that build their Jacobian themselves.  The reference's tracker obtains the same numbers through
        `autograd.grad(sdf / std.detach(), xyz)` (tracker.py:186-192), which works here too (`get_sdf` on a tensor that requires grad), but
        torch's autograd engine adds ~0.2 ms of host time per call to a 0.1 ms kernel; dividing this gradient by `std` gives that result.
        :return: sdf (M,), std (M,), valid_mask (N,) bool, d sdf / d xyz (M, 3)"""
        sdf, std, mask, _, grad = self._query(xyz, True)
        return sdf, std, mask, grad

    # ---- extract ----------------------------------------------------------------------------------------------
    def _new_cache_set(self, capacity: int):
        dev = self.device
        return (torch.empty((capacity, 3, 3), dtype=torch.float32, device=dev), torch.empty((capacity,), dtype=torch.long, device=dev),
                torch.empty((capacity, 3), dtype=torch.float32, device=dev), torch.zeros((capacity,), dtype=torch.uint8, device=dev))

    def _ensure_cache(self, capacity: int):
        if self._cache is not None and self._cache[0].size(0) >= capacity:
            return
        new = self._new_cache_set(capacity)
        if self._cache is not None:
            n = self._cache[0].size(0)
            for a, b in zip(new, self._cache):
                a[:n] = b
        self._cache = new
        self._cache_out = None

    def _cache_clear(self):
        self._counters[_lib.C_CACHE_T] = 0
        self._counters[_lib.C_CACHE_KEPT] = 0
        self._counters[_lib.C_CACHE_DEAD] = 0
        self._tri_n.zero_()
        self._cache_any = False

    def _cache_struct(self):
        b = _lib.DifExtractBuffers()
        b.cache_capacity = self._cache[0].size(0)
        b.cache_tri, b.cache_id, b.cache_std, b.cache_alive = (_lib.ptr(t) for t in self._cache)
        return b

    def _cache_compact(self):
        """Live log entries, in log order, into `_cache_out` (device).  Returns the number of live triangles."""
        cap = self._cache[0].size(0)
        if self._cache_out is None:
            self._cache_out = self._new_cache_set(cap)
        with torch.cuda.device(self.device):
            scratch = torch.empty((4096,), dtype=torch.int32, device=self.device)
            b = self._cache_struct()
            o = self._cache_out
            _lib.check(_lib.load().dif_mesh_cache_compact(ctypes.byref(self._cmap), ctypes.byref(b), _lib.ptr(o[0]), _lib.ptr(o[1]), _lib.ptr(o[2]), cap,
                                                          _lib.ptr(scratch), _lib.stream_ptr()), "dif_mesh_cache_compact")
        return self._read_counters()["cache_live"]

    def _cache_gc(self):
        """Drop the dead entries: the compacted copy becomes the log (same content and order as before for the live part)."""
        with torch.cuda.device(self.device):
            # a deferred triangle export (dif_map_t.pending_export) names ABSOLUTE log rows: it is carried out, in stream order, before
            # the compaction moves them — whichever stepping mode enqueued it and whichever one reaches this safe point
            _lib.check(_lib.load().dif_export_pending(ctypes.byref(self._cmap), _lib.stream_ptr()), "dif_export_pending")
        n = self._cache_compact()
        with torch.cuda.device(self.device):
            # copied back rather than swapped in: the log keeps its addresses, so launch graphs captured over it stay valid
            for dst, src in zip(self._cache[:3], self._cache_out[:3]):
                dst[:n].copy_(src[:n])
            b = self._cache_struct()
            _lib.check(_lib.load().dif_mesh_cache_reindex(ctypes.byref(self._cmap), ctypes.byref(b), n, _lib.stream_ptr()), "dif_mesh_cache_reindex")
        self._gc_epoch += 1
        self._gc_log_len = n
        self._gc_wanted = False
        self._read_counters()
        # the reference's host cache grows without bound (map.py:703-714): if even the compacted log leaves no room for two more
        # calls' worth of triangles, double it (the pointers change: captured launch graphs are re-captured by their owner)
        need = n + 2 * self._cache_call_limit
        if need > self._cache[0].size(0):
            self._ensure_cache(_next_pow2(2 * need))

    def mesh_cache_tensors(self, new_only: bool = False):
        """Device views of the mesh cache: (vertices (T,3,3) f32 world units, vertices_flatten_id (T,) i64, vertices_std (T,3)) — the
        reference's `mesh_cache` arrays, same order; `new_only` restricts to the triangles produced by the last extract.
        None before the first extract."""
        if not self._cache_any:
            return None
        if new_only:
            c = self.last_counters
            lo, hi = c["cache_kept"], c["cache_T"]
            tri, tid, tstd, _ = self._cache
            return tri[lo:hi], tid[lo:hi], tstd[lo:hi]
        n = self._cache_compact()
        o = self._cache_out
        return o[0][:n], o[1][:n], o[2][:n]

    MIN_EXTRACT_ROWS = 1 << 15
    EXTRACT_ROWS_FLOOR = 1 << 12        # rows that `extract_buffer_bytes` never takes away

    def _extract_rows_limit(self, resolution: int) -> int:
        """The most rows `extract_buffer_bytes` allows a streaming map (a power of two, at least EXTRACT_ROWS_FLOOR)."""
        per_voxel = ((2 * resolution) ** 3) * 12 + (resolution ** 3) * 8 + 1024 + 64
        rows = self.EXTRACT_ROWS_FLOOR
        while 2 * rows * per_voxel <= self.extract_buffer_bytes:
            rows *= 2
        return rows

    def _extract_rows(self, resolution: int, no_cache: bool = False) -> int:
        """Rows of the per-voxel extract buffers (~7.7 KB per row at resolution 4).
        A streaming map (untiled, more than 4,096 slots, not `no_cache`) cannot overflow them: an extract whose dirty set could need more rows
        than there are DEFERS itself on the device (`k_dirty_scan`, counters[DIF_C_DEFERRED]: nothing changed, dirty set kept), the host sees
        the rows it wanted with that frame's counters, grows the buffers here, and the next extract meshes the accumulated dirty set.  Such maps
        get four times the high-water mark of what a frame has decoded so far (at least MIN_EXTRACT_ROWS) — not the map's CAPACITY, which was
        4 GB for the 524,288-slot map of a 640x480 stream that decodes 1-13 k voxels per frame.  Every other map keeps rows for its whole
        occupancy bound (there the device can only flag an overflow).  Never shrinks (pointers stay stable), never exceeds the capacity."""
        R = 2 * resolution
        per_voxel = (R ** 3) * 12 + (resolution ** 3) * 8 + 1024 + 64
        if not no_cache and not self._tiled and self._capacity > 4096:
            rows = min(_next_pow2(max(self.MIN_EXTRACT_ROWS, 4 * self._extract_high_water, 2 * self._extract_rows_wanted)), self._extract_rows_limit(resolution))
        else:
            rows = _next_pow2(max(self._n_occ_ub, 1024))
            # Room to grow: four times the occupancy bound, at most the map's capacity and at most `extract_buffer_bytes` — re-allocating
            # ~8 KB per voxel every time the occupancy crosses a power of two costs tens of milliseconds in the middle of a stream
            roomy = min(self._capacity, 4 * rows)
            if roomy * per_voxel <= self.extract_buffer_bytes:
                rows = max(rows, roomy)
        rows = min(rows, _next_pow2(self._capacity))
        if self._xbuf is not None and self._xbuf[0][0] == resolution:
            rows = max(rows, self._xbuf[0][1])
        return rows

    def _extract_buffers(self, resolution: int, max_n_triangles: int, max_vox: int = None, no_cache: bool = False):
        R = 2 * resolution
        if max_vox is None:
            max_vox = self._extract_rows(resolution, no_cache)
        key = (resolution, max_vox)
        if self._xbuf is None or self._xbuf[0] != key:
            dev = self.device
            t = dict(valid_blocks=torch.empty((max_vox,), dtype=torch.long, device=dev),
                     occ_slot=torch.empty((max_vox,), dtype=torch.int32, device=dev),
                     low_sdf=torch.empty((max_vox, resolution ** 3), dtype=torch.float32, device=dev),
                     low_std=torch.empty((max_vox, resolution ** 3), dtype=torch.float32, device=dev),
                     cube_sdf=torch.empty((max_vox, R, R, R), dtype=torch.float32, device=dev),
                     cube_std=torch.empty((max_vox, R, R, R), dtype=torch.float32, device=dev),
                     refine_list=torch.empty((max_vox * R ** 3,), dtype=torch.int32, device=dev),
                     tri_count=torch.empty((max_vox,), dtype=torch.int32, device=dev),
                     tri_offset=torch.empty((max_vox,), dtype=torch.int32, device=dev),
                     block_tmp=torch.empty((4096,), dtype=torch.int32, device=dev),
                     chunk_sum=torch.zeros(((max_vox + 255) // 256 + (max_vox + 65535) // 65536,), dtype=torch.int32, device=dev),
                     mc_status=torch.zeros(((max_vox + 3) // 4 + 256,), dtype=torch.int32, device=dev),     # look-back words + 8 ticket counters, 32 words apart
                     fold_table=torch.empty((max_vox, 256), dtype=torch.float32, device=dev))
            self._xbuf = (key, t)
        t = self._xbuf[1]
        # the log must always have room for two calls' worth of output beyond what the host last saw (counters lag one frame
        # in the pipelined driver); it is garbage-collected in extract_mesh_finish before it gets there
        self._cache_call_limit = int(max_n_triangles)
        self._ensure_cache(max(3 * int(max_n_triangles), 1 << 16))
        b = self._cache_struct()
        b.max_voxels = max_vox
        b.max_triangles = int(max_n_triangles)
        for k, v in t.items():
            setattr(b, k, _lib.ptr(v))
        return t, b

    def extract_mesh_enqueue(self, voxel_resolution: int, max_n_triangles: int, fast: bool = True, max_std: float = 2000.0,
                             no_cache: bool = False):
        """Enqueue the body of `do_meshing` (`map.py:624-714`) without waiting for it: decode the dirty neighbourhood, marching
        cubes, mesh-cache merge, then an async copy of the device counters into pinned host memory.  Returns a handle for
        `extract_mesh_finish`.  Lets a streaming caller keep the GPU queue full (the next frame is enqueued while this one runs)."""
        lib = _lib.load()
        if self._gc_wanted and self._cache is not None:
            self._cache_gc()                                      # safe point: at most one (the latest) extract is still pending
        with self.modifying_lock, self._state_lock, torch.cuda.device(self.device):
            # Whatever stream this extract runs on (the meshing thread uses `meshing_stream`), it starts behind the last integrate,
            # which was enqueued completely under the same lock (the reference drains the device instead, map.py:625).
            if self._integrate_done is not None:
                torch.cuda.current_stream().wait_event(self._integrate_done)
            tens, buf = self._extract_buffers(voxel_resolution, max_n_triangles, no_cache=no_cache)
            w = self.model.packed.weights_struct(self.device)
            _lib.check(lib.dif_extract(ctypes.byref(self._cmap), ctypes.byref(w), ctypes.byref(buf), int(voxel_resolution), 1 if fast else 0,
                                       float(max_std), 1 if no_cache else 0, 1, _lib.stream_ptr()), "dif_extract")
            self.mesh_cache.invalidate_host_copy()
            if self._pinned_counters is None:
                self._pinned_counters = [torch.empty((_lib.C_COUNT,), dtype=torch.int32).pin_memory() for _ in range(4)]
            pc = self._pinned_counters[self._pending_seq % 4]
            self._pending_seq += 1
            pc.copy_(self._counters, non_blocking=True)
            self._counters[_lib.C_OVERFLOW:_lib.C_OVERFLOW + 1].zero_()        # handed over with this snapshot (see _publish_counters_locked)
            ev = torch.cuda.Event()
            ev.record()
            return dict(event=ev, counters=pc, epoch=self._gc_epoch, add_total=self._add_total, max_n_triangles=max_n_triangles)

    def extract_mesh_finish(self, handle):
        """Wait for an enqueued extract and publish its counters (`last_counters`).  Returns the device views of the triangles that
        extract produced (vertices (T,3,3), voxel ids (T,), std (T,3)) — valid until the next-but-one extract overwrites the buffer."""
        if "stamp" in handle:       # the extract's last kernel stamps its pinned counter snapshot: no event sits in the queue for this wait
            _lib.spin_until(handle["counters"], _lib.C_STAMP, handle["stamp"], "extract")
        else:
            handle["event"].synchronize()
        c = self._publish_counters(handle["counters"].tolist(), handle["add_total"])
        if c["deferred"] > 0:               # the extract found its buffers too small and changed nothing: the next one runs with more rows
            self._extract_rows_wanted = max(self._extract_rows_wanted, int(c["deferred"]))
            self.n_deferred += 1
            limit = self._extract_rows_limit(self._xbuf[0][0])
            if int(c["deferred"]) > min(limit, _next_pow2(self._capacity)):
                # (growing cannot help: every later extract would defer as well and the dirty set would never be meshed)
                raise RuntimeError(f"libdifusion: this extract needs {int(c['deferred'])} rows of per-voxel buffers, more than `extract_buffer_bytes` = "
                                   f"{self.extract_buffer_bytes >> 20} MB allows ({limit} rows); it was deferred (nothing is lost: the dirty set is kept) — "
                                   "raise the limit and extract again")
        if c["T"] >= handle["max_n_triangles"]:
            logging.warning(f"Warning from marching cube: the max triangle number is too small {c['T']} vs {handle['max_n_triangles']}")
        if c["K"] > 0:
            self._cache_any = True
        cap = self._cache[0].size(0)
        # compaction is a host-synchronous safe point: amortise it over millions of dead entries (they cost 57 B each, HBM is plentiful)
        if c["cache_dead"] > max(1 << 22, c["cache_T"] // 2) or c["cache_T"] + 2 * handle["max_n_triangles"] > cap:
            self._gc_wanted = True                                # performed at the next enqueue (a safe point)
        tri, tid, tstd, _ = self._cache
        lo, hi = c["cache_kept"], c["cache_T"]
        if handle.get("epoch", self._gc_epoch) != self._gc_epoch:
            # the log was compacted after this extract was enqueued: its triangles (all alive, the newest) are now the log's tail
            n_new = hi - lo
            lo, hi = self._gc_log_len - n_new, self._gc_log_len
            c["cache_kept"], c["cache_T"] = lo, hi
        return tri[lo:hi], tid[lo:hi], tstd[lo:hi]

    def extract_mesh_arrays(self, voxel_resolution: int, max_n_triangles: int, fast: bool = True, max_std: float = 2000.0,
                            no_cache: bool = False, to_host: bool = True):
        """Synchronous `do_meshing` (`map.py:624-714`).  Returns the mesh cache arrays (vertices (T,3,3) f32 world units,
        vertices_flatten_id (T,) i64, vertices_std (T,3) f32) as numpy (`to_host`) or as device views; None while the cache is
        empty and nothing was dirty."""
        self.extract_mesh_finish(self.extract_mesh_enqueue(voxel_resolution, max_n_triangles, fast, max_std, no_cache))
        while self.last_counters["deferred"] > 0:      # grow-and-retry: this call hands the whole dirty set's mesh back (map.py:624-714)
            self.extract_mesh_finish(self.extract_mesh_enqueue(voxel_resolution, max_n_triangles, fast, max_std, no_cache))
        if not self._cache_any:
            return None
        if not to_host:
            return self.mesh_cache_tensors()
        mc = self.mesh_cache
        return mc.vertices, mc.vertices_flatten_id, mc.vertices_std

    def _make_mesh_from_cache(self):
        """reference `map.py:521-543` with Open3D optional."""
        if self.mesh_cache.vertices is None:
            vertices = np.zeros((0, 3), dtype=float)
            std = np.zeros((0,), dtype=float)
        else:
            vertices = self.mesh_cache.vertices.reshape((-1, 3)).astype(float)
            std = self.mesh_cache.vertices_std.reshape((-1,)).astype(float)
        triangles = np.arange(vertices.shape[0]).reshape((-1, 3)).astype(np.int32)
        try:
            import open3d as o3d
        except ImportError:
            return Mesh(vertices, triangles, std)
        final_mesh = o3d.geometry.TriangleMesh()
        final_mesh.vertices = o3d.utility.Vector3dVector(vertices)
        final_mesh.triangles = o3d.utility.Vector3iVector(triangles)
        if vertices.shape[0] > 0:
            import matplotlib.cm
            if self.extract_mesh_std_range is not None:
                lo, hi = self.extract_mesh_std_range
                std = np.clip(std, lo, hi)
            else:
                lo, hi = std.min(), std.max()
            final_mesh.vertex_colors = o3d.utility.Vector3dVector(matplotlib.cm.jet((std - lo) / (hi - lo))[:, :3])
        return final_mesh

    def extract_mesh(self, voxel_resolution: int, max_n_triangles: int, fast: bool = True, max_std: float = 2000.0,
                     extract_async: bool = False, no_cache: bool = False, interpolate: bool = True):
        """reference `map.py:581-723` (same return protocol, including the async one)."""
        if not interpolate:
            raise NotImplementedError("only the interpolating marching cubes exists (as in the reference, map.py:693)")
        if self.meshing_thread is not None:
            if not self.meshing_thread.is_alive():
                self.meshing_thread = None
                self.meshing_thread_id = -1
                return self._make_mesh_from_cache()
            elif not extract_async:
                self.meshing_thread.join()
                return self._make_mesh_from_cache()
            else:
                return None

        def do_meshing():
            with torch.cuda.device(self.device), torch.cuda.stream(self.meshing_stream):
                self.extract_mesh_arrays(voxel_resolution, max_n_triangles, fast, max_std, no_cache)     # ordered behind the integrates inside

        if extract_async:
            self.meshing_thread = threading.Thread(target=do_meshing, daemon=True)
            self.meshing_thread.start()
            self.meshing_thread_id = self.meshing_thread.ident
            return None
        self.extract_mesh_arrays(voxel_resolution, max_n_triangles, fast, max_std, no_cache)
        return self._make_mesh_from_cache()

    # ---- multi-GPU (no reference counterpart) ---------------------------------------------
    HALO_LIST_CAP = 4096                    # entries per boundary change list = records of a delta halo message (512 KB)

    def set_ownership(self, x_lo: int, x_hi: int, halo: int = 3):
        """Spatial tiling (SURVEY.md section 8e "C5"): this map owns the voxels with x index in [x_lo, x_hi).  Integrate ignores points
        whose own voxel is farther than `halo` voxels from the slab and never writes a voxel it does not own; only owned voxels are meshed.
        With the boundary layers refreshed after every integrate (`parallel.exchange_halo`), halo = 3 makes every owned voxel
        bit-identical to the single-map result (see the derivation in parallel.py)."""
        self._ownership = (int(x_lo), int(x_hi), int(halo))
        self._cmap.own_x_lo, self._cmap.own_x_hi, self._cmap.halo = self._ownership
        tiled = x_hi > x_lo and (x_lo > 0 or x_hi < self.n_xyz[0])
        if tiled and self._halo_list is None:
            with torch.cuda.device(self.device):
                self._halo_list = torch.zeros((2, self.HALO_LIST_CAP), dtype=torch.int32, device=self.device)
        self._cmap.halo_list = _lib.ptr(self._halo_list if tiled else None)
        self._cmap.halo_list_cap = self.HALO_LIST_CAP if tiled else 0
        self._halo_lists_stale = True
        with torch.cuda.device(self.device):
            self._counters[_lib.C_HALO_L:_lib.C_HALO_TICKET + 1].zero_()
        if not tiled:
            self._recount_dirty()           # (a tiled map's extract counts the dirty flags itself; the totals are kept only for the whole grid)

    @property
    def _tiled(self) -> bool:
        lo, hi, _ = getattr(self, "_ownership", (0, self.n_xyz[0], 0))
        return hi > lo and (lo > 0 or hi < self.n_xyz[0])

    def export_records(self, x_lo: int = None, x_hi: int = None, raw: bool = False) -> torch.Tensor:
        """(n, 32) int32 records of the allocated voxels with x index in [x_lo, x_hi) (default: all), slot order:
        lin id (2 words) | w | w*z[29]  (raw=False, for additive merging)  or  z[29] itself (raw=True, exact copies)."""
        x_lo = 0 if x_lo is None else max(0, int(x_lo))
        x_hi = self.n_xyz[0] if x_hi is None else min(self.n_xyz[0], int(x_hi))
        cap = max(self._n_occ_ub, 1)
        rec = torch.empty((cap, 32), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            scratch = torch.empty((4096,), dtype=torch.int32, device=self.device)
            _lib.check(_lib.load().dif_export_records(ctypes.byref(self._cmap), _lib.ptr(rec), cap, x_lo, x_hi, 1 if raw else 0,
                                                      _lib.ptr(scratch), _lib.stream_ptr()), "dif_export_records")
        self._read_counters()
        n = int(self._host_counters[_lib.C_EXPORT_N])
        return rec[:n]

    def halo_message_rows(self, layers: int) -> int:
        """Records a halo message of `layers` x-layers can hold (every voxel of the layers allocated)."""
        return int(layers) * self.n_xyz[1] * self.n_xyz[2]

    def export_halo(self, x_lo: int, x_hi: int, out: torch.Tensor = None) -> torch.Tensor:
        """Fixed-size halo message for the x-layers [x_lo, x_hi): (1 + rows, 32) int32 on the device, row 0 = header (word 0 = number of
        records), then raw (w, z, dirty) records in slot order.  Nothing comes back to the host: the length travels in the message."""
        rows = self.halo_message_rows(max(0, min(self.n_xyz[0], int(x_hi)) - max(0, int(x_lo))))
        with torch.cuda.device(self.device):
            if out is None:
                out = torch.zeros((1 + max(rows, 1), 32), dtype=torch.int32, device=self.device)
            if self._halo_scratch is None:
                self._halo_scratch = torch.empty((4096,), dtype=torch.int32, device=self.device)
            _lib.check(_lib.load().dif_export_halo(ctypes.byref(self._cmap), _lib.ptr(out), out.size(0) - 1, int(x_lo), int(x_hi),
                                                   _lib.ptr(self._halo_scratch), _lib.stream_ptr()), "dif_export_halo")
        return out

    def export_halo_delta(self, out_left: Optional[torch.Tensor], out_right: Optional[torch.Tensor], note: Optional[torch.Tensor] = None):
        """Bounded delta messages for both neighbours in ONE launch: the owned boundary voxels this map allocated or fused since the
        last halo export (`dif_export_halo_delta`).  out_left / out_right: (1 + rows, 32) int32 device buffers (None: no neighbour on that
        side, whose change list is left alone); header word 0 = records, word 1 = records that were pending (> word 0: overflow), word 2 = 1.
        note: optional int32[8] (pinned host memory is fine) receiving both headers."""
        rows = min(t.size(0) - 1 for t in (out_left, out_right) if t is not None)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().dif_export_halo_delta(ctypes.byref(self._cmap), _lib.ptr(out_left), _lib.ptr(out_right), min(rows, self.HALO_LIST_CAP),
                                                         _lib.ptr(note), _lib.stream_ptr()), "dif_export_halo_delta")

    def halo_lists_reset(self, hdr_left: Optional[torch.Tensor], hdr_right: Optional[torch.Tensor], note: Optional[torch.Tensor] = None):
        """After whole-layer messages (which supersede the change lists): complete their headers (word 1 = what a delta would have carried,
        word 2 = 0) and empty the lists."""
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().dif_halo_lists_reset(ctypes.byref(self._cmap), _lib.ptr(hdr_left), _lib.ptr(hdr_right), _lib.ptr(note), _lib.stream_ptr()),
                       "dif_halo_lists_reset")

    def merge_halo(self, msg: torch.Tensor, rows: int = None):
        """Overwrite (w, z, dirty) of the voxels named in a halo message (allocating the unseen ones, ascending id); the record count is
        read on the device.  `rows`: records the message can hold (default: its whole length)."""
        self.merge_halo2(msg, None, rows, None)

    def merge_halo2(self, msg_a: Optional[torch.Tensor], msg_b: Optional[torch.Tensor], rows_a: int = None, rows_b: int = None,
                    note: Optional[torch.Tensor] = None, reserved: bool = False):
        """`merge_halo` for the messages of both neighbours in one pass (`dif_merge_halo2`); note: optional int32[8] receiving the two
        headers as received.  reserved: the caller has already made room for the voxels the messages may allocate (`_ensure_capacity`)."""
        rows = []
        for m, r in ((msg_a, rows_a), (msg_b, rows_b)):
            if m is not None:
                _lib.require_cuda(m)
            rows.append(0 if m is None else (m.size(0) - 1 if r is None else min(int(r), m.size(0) - 1)))
        if rows[0] + rows[1] == 0:
            return
        with self.modifying_lock, self._state_lock, torch.cuda.device(self.device):
            if not reserved:
                self._ensure_capacity(rows[0] + rows[1])
            if self._halo_scratch is None:
                self._halo_scratch = torch.empty((4096,), dtype=torch.int32, device=self.device)
            _lib.check(_lib.load().dif_merge_halo2(ctypes.byref(self._cmap), _lib.ptr(msg_a), rows[0], _lib.ptr(msg_b), rows[1],
                                                   _lib.ptr(self._halo_scratch), _lib.ptr(note), _lib.stream_ptr()), "dif_merge_halo2")
            if not self._tiled:
                self._recount_dirty()       # (a tiled map's extract counts the dirty flags itself)
            if self._integrate_done is None:
                self._integrate_done = torch.cuda.Event()
            self._integrate_done.record()

    def merge_records(self, rec: torch.Tensor, assign: bool = False):
        """Fold records with DISTINCT lin ids into this map: accumulate (one rank's `export_records()`), or `assign`
        (overwrite w and z with `export_records(raw=True)` payloads: halo refresh)."""
        rec = rec.contiguous()
        n = rec.size(0)
        if n == 0:
            return
        with self.modifying_lock, self._state_lock, torch.cuda.device(self.device):
            self._ensure_capacity(n)
            scratch = torch.empty((4096,), dtype=torch.int32, device=self.device)
            _lib.check(_lib.load().dif_merge_records(ctypes.byref(self._cmap), _lib.ptr(rec), n, 1 if assign else 0, _lib.ptr(scratch),
                                                     _lib.stream_ptr()), "dif_merge_records")
            self._halo_lists_stale = True
            self._recount_dirty()
            if self._integrate_done is None:
                self._integrate_done = torch.cuda.Event()
            self._integrate_done.record()                          # writes latents like an integrate: later extracts wait for it

    # ---- visualisers: out of scope (need Open3D) ---------------------------------------
    def get_fast_preview_visuals(self):
        raise NotImplementedError("Open3D visualisers are outside the fusion hot path")

    def get_map_visuals(self, *a, **k):
        raise NotImplementedError("Open3D visualisers are outside the fusion hot path")
