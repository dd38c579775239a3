"""ctypes binding of libdifusion.so (the C ABI declared in include/difusion.h).

There is no CPU fallback: every compute entry point of this package goes through this library and raises
`RuntimeError` if it is missing or a call fails.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_float, c_int32, c_int64, c_uint8, c_void_p
from pathlib import Path

PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ["DIF_LIB"]) if os.environ.get("DIF_LIB") else PKG / "libdifusion.so"    # DIF_LIB: instrumented builds (tools/)

# counters (difusion.h)
C_N_OCCUPIED, C_OVERFLOW, C_ALLOC_NEW, C_M, C_C, C_ITEMS, C_K, C_B, C_VH, C_T, C_QUERY_M, C_N_KEPT, C_CACHE_T, C_CACHE_KEPT, C_EXPORT_N, C_WORK, C_CACHE_DEAD, C_CACHE_LIVE, C_OPT_ROWS, C_OPT_VOXELS, C_HALO_L, C_HALO_R, C_HALO_TICKET, C_DEFERRED = range(24)
C_STAMP = 31
SYNC_FUSED, SYNC_FRONT_DONE, SYNC_WORDS = 0, 32, 64      # dif_map_t.sync_words
FC_COUNT = 32                                            # dif_map_t.frame_counters
C_COUNT = 32
PROF_NAMES = ["encode", "decode_lattice", "decode_points", "mc_count", "mc_emit", "halo_export", "halo_merge"]
PROF_COUNT = 8
LATENT_DIM = 29

ERRORS = {-1: "DIF_EINVAL (bad argument)", -2: "DIF_ELAUNCH (HIP launch/runtime failure)", -3: "DIF_ENOSPACE (workspace too small)"}


class DifMap(Structure):
    _fields_ = [("nx", c_int32), ("ny", c_int32), ("nz", c_int32),
                ("bound_min", c_float * 3), ("voxel_size", c_float),
                ("prune_min_vox_obs", c_int32), ("ignore_count_th", c_float), ("encoder_count_th", c_float),
                ("capacity", c_int64),
                ("indexer", c_void_p), ("latent_vecs", c_void_p), ("latent_vecs_pos", c_void_p),
                ("voxel_obs_count", c_void_p), ("dirty", c_void_p), ("voxel_optimized", c_void_p), ("counters", c_void_p),
                ("frame_count", c_void_p), ("grid_bits", c_void_p), ("grid_tot", c_void_p), ("vbm", c_void_p),
                ("rec_dir", c_void_p), ("upd_list", c_void_p),
                ("tri_start", c_void_p), ("tri_n", c_void_p),
                ("own_x_lo", c_int32), ("own_x_hi", c_int32), ("halo", c_int32), ("dirty_tot", c_void_p),
                ("halo_list", c_void_p), ("halo_list_cap", c_int32), ("pending_export", c_void_p),
                ("alloc_bits", c_void_p), ("alloc_tot", c_void_p), ("sync_words", c_void_p), ("frame_seq", c_int32), ("fuse_stream", c_void_p), ("frame_counters", c_void_p)]


class DifWeights(Structure):
    _fields_ = [("enc_packed", c_void_p), ("enc_packed_floats", c_int64),
                ("dec_packed", c_void_p), ("dec_packed_floats", c_int64),
                ("dec_bwd_packed", c_void_p), ("dec_bwd_packed_floats", c_int64),
                ("dec_fold_packed", c_void_p), ("dec_fold_packed_floats", c_int64),
                ("dec_x6_packed", c_void_p), ("dec_x6_packed_bytes", c_int64),
                ("enc_x6_packed", c_void_p), ("enc_x6_packed_bytes", c_int64),
                ("dec_x6u_packed", c_void_p), ("dec_x6u_packed_bytes", c_int64),
                ("dec_x6b_packed", c_void_p), ("dec_x6b_packed_bytes", c_int64)]


class DifExtractBuffers(Structure):
    _fields_ = [("max_voxels", c_int64), ("valid_blocks", c_void_p), ("occ_slot", c_void_p),
                ("low_sdf", c_void_p), ("low_std", c_void_p), ("cube_sdf", c_void_p), ("cube_std", c_void_p),
                ("refine_list", c_void_p), ("tri_count", c_void_p), ("tri_offset", c_void_p), ("block_tmp", c_void_p),
                ("max_triangles", c_int64), ("cache_capacity", c_int64),
                ("cache_tri", c_void_p), ("cache_id", c_void_p), ("cache_std", c_void_p), ("cache_alive", c_void_p),
                ("counters_out", c_void_p), ("out_tri", c_void_p), ("out_id", c_void_p), ("out_std", c_void_p), ("out_capacity", c_int64),
                ("chunk_sum", c_void_p), ("fold_table", c_void_p), ("mc_status", c_void_p), ("defer_export", c_int32),
                ("stamp", c_int32), ("export_notify", c_void_p)]


MAX_STREAMS = 8          # DIF_MAX_STREAMS


class DifStreamFrame(Structure):
    """dif_stream_frame_t: one stream's share of a batched frame (dif_integrate_frames / dif_extract_streams)."""
    _fields_ = [("map", POINTER(DifMap)), ("frame_dev", c_void_p), ("xyz_world", c_void_p), ("normal_world", c_void_p), ("unq_mask", c_void_p),
                ("ws", c_void_p), ("ws_bytes", c_int64), ("buf", POINTER(DifExtractBuffers))]


# name -> (restype, argtypes); mirrors include/difusion.h one to one (tests/test_abi.py checks the symbol list)
class DifSdfHg(ctypes.Structure):
    """include/difusion.h: dif_sdf_hg_t"""
    _fields_ = [("T_cur", c_float * 12), ("T_delta", c_float * 12), ("last_Rt", c_float * 9), ("robust_kernel", c_int32), ("robust_k", c_float),
                ("no_grad", c_int32)]


SIGNATURES = {
    "dif_version": (c_int32, []),
    "dif_build_id": (ctypes.c_char_p, []),
    "dif_unproject": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_float, c_float, c_float, c_float, c_void_p]),
    "dif_unproject_transform": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_float, c_float,
                                          c_float, c_float, POINTER(c_float), POINTER(c_float), c_void_p]),
    "dif_unproject_transform_dev": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_float, c_float,
                                              c_float, c_float, c_void_p, c_void_p]),
    "dif_unproject_transform_frame": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_float, c_float, c_float, c_float, c_void_p]),
    "dif_compute_normal_weight": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "dif_filter_depth": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "dif_depth_frontend": (c_int32, [c_void_p, c_int32, c_int32, c_float, c_float, c_float, c_float, c_int32, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p]),
    "dif_point_box_filter": (c_int32, [c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                       c_void_p, c_void_p, c_void_p]),
    "dif_cloud_workspace_bytes": (c_int64, [c_int64]),
    "dif_knn": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "dif_remove_radius_outlier": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_float, c_void_p, c_void_p, c_int64, c_void_p]),
    "dif_estimate_normals": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_float, POINTER(c_float), c_void_p, c_void_p, c_int64,
                                       c_void_p]),
    "dif_groupby_sum": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_int64, c_void_p]),
    "dif_integrate_workspace_bytes": (c_int64, [c_int64]),
    "dif_integrate": (c_int32, [POINTER(DifMap), POINTER(DifWeights), c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                c_int64, c_void_p]),
    "dif_integrate_frame": (c_int32, [POINTER(DifMap), POINTER(DifWeights), c_void_p, c_int32, c_int32, c_float, c_float, c_float, c_float,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "dif_optimize_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "dif_optimize_latents": (c_int32, [POINTER(DifMap), POINTER(DifWeights), c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_float,
                                       c_float, c_void_p, c_void_p, c_int64, c_void_p]),
    "dif_extract": (c_int32, [POINTER(DifMap), POINTER(DifWeights), POINTER(DifExtractBuffers), c_int32, c_int32, c_float,
                              c_int32, c_int32, c_void_p]),
    "dif_integrate_frames": (c_int32, [POINTER(DifStreamFrame), c_int32, POINTER(DifWeights), c_int32, c_int32, c_float, c_float, c_float, c_float,
                                       c_void_p]),
    "dif_extract_streams": (c_int32, [POINTER(DifStreamFrame), c_int32, POINTER(DifWeights), c_int32, c_float, c_int32, c_void_p]),
    "dif_export_pending": (c_int32, [POINTER(DifMap), c_void_p]),
    "dif_mesh_cache_export": (c_int32, [POINTER(DifExtractBuffers), c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dif_mesh_cache_export_dma": (c_int32, [POINTER(DifExtractBuffers), c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dif_mesh_cache_compact": (c_int32, [POINTER(DifMap), POINTER(DifExtractBuffers), c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "dif_mesh_cache_reindex": (c_int32, [POINTER(DifMap), POINTER(DifExtractBuffers), c_int64, c_void_p]),
    "dif_marching_cubes": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                     c_void_p, c_int32, c_float, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p]),
    "dif_decode_rows": (c_int32, [POINTER(DifWeights), c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "dif_encode_rows": (c_int32, [POINTER(DifWeights), c_void_p, c_int64, c_void_p, c_void_p]),
    "dif_query_sdf": (c_int32, [POINTER(DifMap), POINTER(DifWeights), c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_void_p]),
    "dif_query_select": (c_int32, [POINTER(DifMap), c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "dif_query_decode": (c_int32, [POINTER(DifMap), POINTER(DifWeights), c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dif_query_grad_scatter": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "dif_query_grad_gather": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "dif_export_records": (c_int32, [POINTER(DifMap), c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "dif_merge_records": (c_int32, [POINTER(DifMap), c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "dif_export_halo": (c_int32, [POINTER(DifMap), c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "dif_merge_halo": (c_int32, [POINTER(DifMap), c_void_p, c_int64, c_void_p, c_void_p]),
    "dif_export_halo_delta": (c_int32, [POINTER(DifMap), c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "dif_halo_lists_reset": (c_int32, [POINTER(DifMap), c_void_p, c_void_p, c_void_p, c_void_p]),
    "dif_merge_halo2": (c_int32, [POINTER(DifMap), c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "dif_profile_enable": (c_int32, [c_int32]),
    "dif_profile_read": (c_int32, [POINTER(ctypes.c_double), POINTER(c_int64), c_int32]),
    "dif_profile_dump": (c_int64, [POINTER(c_int32), POINTER(c_float), c_int64, c_int32]),
    "dif_read_counters": (c_int32, [POINTER(DifMap), POINTER(c_int32), c_void_p]),
    "dif_test_mc_grid_cap": (c_int32, [c_int32]),
    "dif_test_sdma_mode": (c_int32, [c_int32]),
    "dif_sdma_info": (c_int32, [POINTER(c_int32)]),
    "dif_test_handoff": (c_int32, [c_int32, c_int32, c_int32, c_int32, POINTER(c_int64)]),
    "dif_queues_independent": (c_int32, [c_void_p, c_void_p]),
    "dif_mesh_cache_export_sdma": (c_int32, [POINTER(DifExtractBuffers), c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "dif_sdf_hg_workspace_bytes": (c_int64, [c_int64]),
    "dif_sdf_hg": (c_int32, [POINTER(DifMap), POINTER(DifWeights), c_void_p, c_int64, POINTER(DifSdfHg), c_void_p, c_int64, c_void_p, c_void_p,
                             c_int64, c_void_p]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load libdifusion.so; loud failure when it has not been built (`python __graft_entry__.py build`)."""
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (torch's copy of the HIP runtime must be the one in the process before the library binds to it)
        if not LIB_PATH.exists():
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc, gfx950). di_fusion_amd has no CPU fallback.")
        if not os.environ.get("DIF_LIB"):
            # provenance: the library carries a hash of the sources it was built from; a library from another tree (stale after a checkout,
            # an rsync, an edit without a rebuild) is rebuilt here rather than silently used
            from . import _build
            have, want = _build.lib_build_id(), _build.source_hash()
            if have != want:
                try:
                    _build.build(force=True, verbose=False, shared=True)      # (processes racing for the same stale library share one build)
                except Exception as e:
                    raise RuntimeError(f"{LIB_PATH} was built from other sources (build id {have}, tree {want}) and rebuilding it failed: {e!r}")
        lib = ctypes.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def spin_until(words, index: int, value: int, what: str, timeout_s: float = 30.0):
    """Busy-wait until `words[index] == value` — a word of pinned host memory that a kernel writes when it is done (a completion signal
    that needs no event in the HIP queue: an event record costs the GPU ~5 us between two kernels).  Raises after `timeout_s`."""
    import time
    if words[index] == value:
        return
    t0 = time.perf_counter()
    n = 0
    while words[index] != value:
        n += 1
        if (n & 0x3FFF) == 0 and time.perf_counter() - t0 > timeout_s:
            raise RuntimeError(f"libdifusion: {what} did not complete within {timeout_s:.0f} s (expected stamp {value}, have {int(words[index])})")


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"libdifusion: {what} failed: {ERRORS.get(rc, rc)}")


def ptr(t) -> c_void_p:
    """Device (or host) pointer of a torch tensor; None -> NULL."""
    if t is None:
        return c_void_p(0)
    return c_void_p(t.data_ptr())


def stream_ptr() -> c_void_p:
    """The current HIP stream of the current device as a raw pointer (what every entry point takes)."""
    import torch
    try:        # ~1 us; `torch.cuda.current_stream().cuda_stream` builds a Stream object and costs ~8 us — six of them per frame add up
        return c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))
    except AttributeError:
        return c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("di_fusion_amd: tensors must live on the GPU (no CPU fallback); got a CPU tensor")
        if t is not None and not t.is_contiguous():
            raise RuntimeError("di_fusion_amd: tensors must be contiguous")
