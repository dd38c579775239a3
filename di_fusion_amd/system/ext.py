"""Flat operator API of the reference's `pytorch/system/ext/__init__.py:15-44`, bound to libdifusion.so.

Same names, argument order and error behaviour (RuntimeError on non-GPU / non-contiguous input, like the
`CHECK_INPUT` macros of the reference extensions); outputs are allocated here as torch tensors on the input's device.
"""
from __future__ import annotations

from typing import List

import torch

from .. import _lib


def _dev(t):
    return torch.cuda.device(t.device)


def unproject_depth(depth: torch.Tensor, fx: float, fy: float, cx: float, cy: float) -> torch.Tensor:
    """(H,W) f32 -> (H,W,3) f32.  reference `ext/imgproc/imgproc.cu:26-44`.  NaN depth -> NaN point (all three)."""
    _lib.require_cuda(depth)
    H, W = depth.shape
    pc = torch.empty((H, W, 3), dtype=torch.float32, device=depth.device)
    with _dev(depth):
        _lib.check(_lib.load().dif_unproject(_lib.ptr(depth), _lib.ptr(pc), H, W, fx, fy, cx, cy, _lib.stream_ptr()), "dif_unproject")
    return pc


def compute_normal_weight(pc_map: torch.Tensor) -> torch.Tensor:
    """(H,W,3) -> (H,W,4) normal + weight, w=-1 invalid.  reference `ext/imgproc/imgproc.cu:143-160`."""
    _lib.require_cuda(pc_map)
    H, W, _ = pc_map.shape
    out = torch.empty((H, W, 4), dtype=torch.float32, device=pc_map.device)
    with _dev(pc_map):
        _lib.check(_lib.load().dif_compute_normal_weight(_lib.ptr(pc_map), _lib.ptr(out), H, W, _lib.stream_ptr()), "dif_compute_normal_weight")
    return out


def groupby_sum(values: torch.Tensor, indices: torch.Tensor, C) -> List[torch.Tensor]:
    """[sum (C,L) f32, count (C,) i32].  reference `ext/indexing/indexing.cu:89-109` (counts here are per sample; the
    reference's kernel counts L per sample, a bug its only caller never observes)."""
    _lib.require_cuda(values, indices)
    C = int(C)
    N, L = values.shape
    s = torch.zeros((C, L), dtype=torch.float32, device=values.device)
    c = torch.zeros((C,), dtype=torch.int32, device=values.device)
    with _dev(values):
        _lib.check(_lib.load().dif_groupby_sum(_lib.ptr(values), _lib.ptr(indices), N, L, _lib.ptr(s), _lib.ptr(c), C, _lib.stream_ptr()), "dif_groupby_sum")
    return [s, c]


def depth_frontend(depth: torch.Tensor, fx: float, fy: float, cx: float, cy: float, filter: bool = True, want_frame: bool = False):
    """`filter_depth` -> `unproject_depth` -> `compute_normal_weight` (reference `ext/imgproc/imgproc.cu:48-160`) in one pass over 16 x 16
    pixel tiles (depth tile + apron in LDS), bit-identical to the three calls.  Returns (depth_filtered (H,W), pc (H,W,3),
    normal_weight (H,W,4)) and, with `want_frame`, also (frame_depth (H,W), frame_normal (H,W,3)): NaN where no normal exists — the
    inputs `dif_integrate_frame` takes, for streams that come without normals."""
    _lib.require_cuda(depth)
    H, W = depth.shape
    dev = depth.device
    d = torch.empty((H, W), dtype=torch.float32, device=dev)
    pc = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
    nw = torch.empty((H, W, 4), dtype=torch.float32, device=dev)
    fd = torch.empty((H, W), dtype=torch.float32, device=dev) if want_frame else None
    fn = torch.empty((H, W, 3), dtype=torch.float32, device=dev) if want_frame else None
    with _dev(depth):
        _lib.check(_lib.load().dif_depth_frontend(_lib.ptr(depth), H, W, fx, fy, cx, cy, 1 if filter else 0, _lib.ptr(d), _lib.ptr(pc), _lib.ptr(nw),
                                                  _lib.ptr(fd), _lib.ptr(fn), _lib.stream_ptr()), "dif_depth_frontend")
    return (d, pc, nw, fd, fn) if want_frame else (d, pc, nw)


def pack_batch(indices: torch.Tensor, n_batch: int, n_point: int):
    """reference `ext/indexing/indexing.cu:73-86`; only reachable through `pack_samples`, which nothing on the fusion
    path calls."""
    raise NotImplementedError("pack_batch is off the fusion path (dead in the reference product)")


def marching_cubes_interp(indexer: torch.Tensor, valid_blocks: torch.Tensor, vec_batch_mapping: torch.Tensor,
                          cube_sdf: torch.Tensor, cube_std: torch.Tensor, max_n_triangles: int, n_xyz, max_std: float):
    """[triangles (T,3,3) f32, triangle_flatten_id (T,) i64, triangle_std (T,3) f32] in voxel units.
    reference `ext/marching_cubes/mc.cpp:3-16`, `mc_interp_kernel.cu:322-382`.  Output order is canonical
    (voxel, cell, table order) instead of atomic arrival order."""
    _lib.require_cuda(indexer, valid_blocks, vec_batch_mapping, cube_sdf, cube_std)
    dev = indexer.device
    K = valid_blocks.size(0)
    R = cube_sdf.size(1) if cube_sdf.dim() == 4 and cube_sdf.size(0) > 0 else 2
    r3 = max(1, (R // 2) ** 3)
    cap = int(min(max_n_triangles, max(1, K * r3 * 5)))
    tri = torch.empty((cap, 3, 3), dtype=torch.float32, device=dev)
    tid = torch.empty((cap,), dtype=torch.int64, device=dev)
    tstd = torch.empty((cap, 3), dtype=torch.float32, device=dev)
    cnt = torch.empty((max(K, 1),), dtype=torch.int32, device=dev)
    off = torch.empty((max(K, 1),), dtype=torch.int32, device=dev)
    tmp = torch.empty((4096,), dtype=torch.int32, device=dev)
    counters = torch.zeros((_lib.C_COUNT,), dtype=torch.int32, device=dev)
    with _dev(indexer):
        _lib.check(_lib.load().dif_marching_cubes(_lib.ptr(indexer), int(n_xyz[0]), int(n_xyz[1]), int(n_xyz[2]), _lib.ptr(valid_blocks), K,
                                                  _lib.ptr(vec_batch_mapping), vec_batch_mapping.size(0), _lib.ptr(cube_sdf), _lib.ptr(cube_std),
                                                  R, float(max_std), cap, _lib.ptr(tri), _lib.ptr(tid), _lib.ptr(tstd), _lib.ptr(cnt),
                                                  _lib.ptr(off), _lib.ptr(tmp), _lib.ptr(counters), _lib.stream_ptr()), "dif_marching_cubes")
    T = int(counters[_lib.C_T].item())
    if T >= max_n_triangles and T > cap:
        import sys
        print(f"Warning from marching cube: the max triangle number is too small {T} vs {max_n_triangles}", file=sys.stderr)
    T = min(T, cap)
    return [tri[:T], tid[:T], tstd[:T]]


def filter_depth(depth_in: torch.Tensor, depth_out: torch.Tensor) -> None:
    """In-place-style bilateral depth filter, reference `ext/imgproc/imgproc.cu:81-94` (writes `depth_out`, border untouched)."""
    _lib.require_cuda(depth_in, depth_out)
    H, W = depth_in.shape
    with _dev(depth_in):
        _lib.check(_lib.load().dif_filter_depth(_lib.ptr(depth_in), _lib.ptr(depth_out), H, W, _lib.stream_ptr()), "dif_filter_depth")


_PBF_SCRATCH = {}


def point_box_filter(points: torch.Tensor, normals: torch.Tensor, voxel_size: float, max_cells: int = 1 << 27):
    """Voxel-mean down-sampling, reference `system/tracker.py:13-23` (the step right before `integrate_keyframe`):
    (N,3) points + normals -> (n_boxes,3) means, boxes ordered by ascending linear box id.  `max_cells` bounds the box grid
    (bounding box of the cloud / voxel_size + 16 per axis); 2^27 cells = 16 MB of bitmap."""
    _lib.require_cuda(points, normals)
    N = points.size(0)
    dev = points.device
    key = (str(dev), int(max_cells))
    with _dev(points):
        if key not in _PBF_SCRATCH:
            nw = (max_cells + 31) // 32
            _PBF_SCRATCH[key] = (torch.zeros((nw,), dtype=torch.int32, device=dev), torch.empty((nw,), dtype=torch.int32, device=dev))
        bits, word_rank = _PBF_SCRATCH[key]
        out_p = torch.empty((max(N, 1), 3), dtype=torch.float32, device=dev)
        out_n = torch.empty((max(N, 1), 3), dtype=torch.float32, device=dev)
        cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
        sums = torch.empty((max(N, 1) * 8,), dtype=torch.int64, device=dev)
        scratch = torch.zeros((4200,), dtype=torch.int32, device=dev)
        _lib.check(_lib.load().dif_point_box_filter(_lib.ptr(points), _lib.ptr(normals), N, float(voxel_size), _lib.ptr(out_p), _lib.ptr(out_n),
                                                    _lib.ptr(cnt), _lib.ptr(bits), int(max_cells), _lib.ptr(word_rank), _lib.ptr(sums),
                                                    _lib.ptr(scratch), _lib.stream_ptr()), "dif_point_box_filter")
        st = scratch[4102].item()
    if st != 0:
        raise RuntimeError("point_box_filter: the cloud's box grid exceeds max_cells")
    n = int(cnt.item())
    return out_p[:n], out_n[:n]


# ---- ext/pcproc (tracker.py:105-113) -----------------------------------------------------------------
_CLOUD_WS = {}


def _cloud_workspace(dev, n: int) -> torch.Tensor:
    need = int(_lib.load().dif_cloud_workspace_bytes(max(int(n), 1)))
    if need < 0:
        raise RuntimeError("point cloud too large for the neighbourhood index")
    ws = _CLOUD_WS.get(str(dev))
    if ws is None or ws.numel() < need:
        ws = torch.empty((int(need * 1.25),), dtype=torch.uint8, device=dev)
        _CLOUD_WS[str(dev)] = ws
    return ws


def _cloud_input(input_pc: torch.Tensor):
    _lib.require_cuda(input_pc)
    if input_pc.dim() != 2 or input_pc.size(1) not in (3, 4) or input_pc.dtype != torch.float32:
        raise RuntimeError("input_pc must be a (N,3) or (N,4) float32 tensor")
    return int(input_pc.size(0)), int(input_pc.size(1))


def knn_search(input_pc: torch.Tensor, k: int, radius: float):
    """Exact k nearest neighbours of every point inside its own cloud, bounded by `radius`: (idx (N,k) int32, dist2 (N,k) f32),
    ascending by (dist2, idx), self included; entries at or beyond the radius are (-1, inf).  What the reference's
    `KDTreeCuda3dIndex.knnSearch` (`ext/pcproc/cuda_kdtree.cu:1173`) feeds to its two kernels."""
    n, stride = _cloud_input(input_pc)
    idx = torch.empty((n, int(k)), dtype=torch.int32, device=input_pc.device)
    dist = torch.empty((n, int(k)), dtype=torch.float32, device=input_pc.device)
    with _dev(input_pc):
        ws = _cloud_workspace(input_pc.device, n)
        _lib.check(_lib.load().dif_knn(_lib.ptr(input_pc), n, stride, int(k), float(radius), _lib.ptr(idx), _lib.ptr(dist), _lib.ptr(ws),
                                       ws.numel(), _lib.stream_ptr()), "dif_knn")
    return idx, dist


def remove_radius_outlier(input_pc: torch.Tensor, nb_points: int, radius: float) -> torch.Tensor:
    """(N,) bool: the point has at least `nb_points` points (itself included) closer than `radius`.
    reference `ext/pcproc/pcproc.cu:160-186`."""
    n, stride = _cloud_input(input_pc)
    mask = torch.empty((n,), dtype=torch.bool, device=input_pc.device)
    with _dev(input_pc):
        ws = _cloud_workspace(input_pc.device, n)
        _lib.check(_lib.load().dif_remove_radius_outlier(_lib.ptr(input_pc), n, stride, int(nb_points), float(radius), _lib.ptr(mask),
                                                         _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "dif_remove_radius_outlier")
    return mask


def estimate_normals(input_pc: torch.Tensor, max_nn: int, radius: float, cam_xyz) -> torch.Tensor:
    """(N,3) PCA normals over the `max_nn` nearest neighbours inside `radius`, oriented towards `cam_xyz`; NaN rows where
    fewer than 5 neighbours qualify.  reference `ext/pcproc/pcproc.cu:188-209`."""
    import ctypes
    n, stride = _cloud_input(input_pc)
    out = torch.empty((n, 3), dtype=torch.float32, device=input_pc.device)
    cam = (ctypes.c_float * 3)(float(cam_xyz[0]), float(cam_xyz[1]), float(cam_xyz[2]))
    with _dev(input_pc):
        ws = _cloud_workspace(input_pc.device, n)
        _lib.check(_lib.load().dif_estimate_normals(_lib.ptr(input_pc), n, stride, int(max_nn), float(radius), cam, _lib.ptr(out), _lib.ptr(ws),
                                                    ws.numel(), _lib.stream_ptr()), "dif_estimate_normals")
    return out
