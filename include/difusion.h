/*
 * difusion.h — C ABI of libdifusion.so, the MI355X (gfx950) implementation of DI-Fusion's per-frame fusion path.
 *
 * This is the drop-in boundary: the entry points below are what the reference's four pybind11 torch extensions
 * (`/root/reference/pytorch/system/ext/__init__.py:15-44`) plus the torch-op sequences of
 * `pytorch/system/map.py` (integrate_keyframe :340-519, extract_mesh :581-723, get_sdf :559-579) bind to.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless marked [host] (a few outputs and the frame descriptor
 *     may also live in device-mapped pinned host memory: said where it applies);
 *   - all buffers are caller-owned (the Python façade allocates them as torch tensors); no hidden allocation;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued, nothing synchronises except dif_read_counters();
 *   - return 0 on success, a negative DIF_E* code on a bad argument / launch failure (no exceptions, no exit);
 *   - element counts that are only known on the device (#voxels allocated, #rows gathered, #triangles) stay on the
 *     device in the map's `counters` array (DIF_C_*); kernels read them there, the host reads them only at the end of extract.
 *
 * Linear voxel id:  lin = z + nz*y + nz*ny*x  of  ceil((p - bound_min)/voxel_size) - 1   (map.py:287-292,366-369).
 */
#ifndef DIFUSION_H
#define DIFUSION_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIF_VERSION 100          /* 0.1.0 */
#define DIF_LATENT_DIM 29        /* ckpt/default/hyper.json:34 */

#define DIF_MAX_STREAMS 8        /* independent maps one batched launch chain can carry (dif_integrate_frames / dif_extract_streams) */

#define DIF_OK 0
#define DIF_EINVAL (-1)
#define DIF_ELAUNCH (-2)
#define DIF_ENOSPACE (-3)

/* Device-resident counters (one per map).  int32 each; indices into the `counters` array. */
enum {
    DIF_C_N_OCCUPIED = 0,   /* map.py:200  n_occupied                                                   */
    DIF_C_OVERFLOW = 1,     /* set !=0 when a device-side buffer was too small (checked by the façade); 7 = the one-pass marching cubes gave up a look-back;
                             * 8 = a delta halo message / boundary change list overflowed */
    DIF_C_ALLOC_NEW = 2,    /* voxels allocated by the last integrate                                     */
    DIF_C_M = 3,            /* gathered (point, offset) rows of the last integrate  (map.py:434-435)      */
    DIF_C_C = 4,            /* voxels updated by the encoder in the last integrate  (map.py:437)          */
    DIF_C_ITEMS = 5,        /* encoder tiles (32 gathered rows each) of the last integrate                */
    DIF_C_K = 6,            /* dirty voxels handed to marching cubes ("valid_blocks", map.py:627)         */
    DIF_C_B = 7,            /* confident voxels decoded ("occupied_vec_id", map.py:631)                   */
    DIF_C_VH = 8,           /* refine rows of the fast two-level decode (map.py:667)                      */
    DIF_C_T = 9,            /* triangles produced (may exceed max_n_triangles, mc_interp_kernel.cu:369)   */
    DIF_C_QUERY_M = 10,     /* valid points of the last get_sdf query (map.py:569-572)                    */
    DIF_C_N_KEPT = 11,      /* points surviving the >prune_min_vox_obs filter (map.py:375)                */
    DIF_C_CACHE_T = 12,     /* entries in the device-resident mesh-cache LOG (live + dead; map.py:116-133, 703-714) */
    DIF_C_CACHE_KEPT = 13,  /* log length before the last extract = offset of that extract's new triangles */
    DIF_C_EXPORT_N = 14,    /* records written by the last dif_export_records                             */
    DIF_C_WORK = 15,        /* scratch                                                                    */
    DIF_C_CACHE_DEAD = 16,  /* dead entries in the mesh-cache log (replaced triangles awaiting compaction)  */
    DIF_C_CACHE_LIVE = 17,  /* triangles written by the last dif_mesh_cache_compact                        */
    DIF_C_OPT_ROWS = 18,    /* samples gathered by the last dif_optimize_latents (n_samples, map.py:85)      */
    DIF_C_OPT_VOXELS = 19,  /* voxels optimised by it (latent_id_subset_uniques, map.py:496)                  */
    DIF_C_HALO_L = 20,      /* entries appended to the LEFT / RIGHT boundary change list (dif_map_t.halo_list) since the last    */
    DIF_C_HALO_R = 21,      /* halo export; may exceed halo_list_cap (then the list is incomplete and the delta export says so)   */
    DIF_C_HALO_TICKET = 22, /* idle 0: workgroups of dif_export_halo_delta that are done                                          */
    DIF_C_DEFERRED = 23,    /* 0, or the rows the per-voxel extract buffers would have needed when the last streaming extract (dirty_tot set, untiled,
                             * capacity > 4096) found them too small for min(7 K, n_occupied) voxels: that extract then changed NOTHING (dirty set kept,
                             * K = B = T = 0) — the caller grows dif_extract_buffers_t.max_voxels and the next extract meshes the accumulated dirty set */
    DIF_C_STAMP = 31,       /* snapshots handed to the caller only (dif_extract_buffers_t.counters_out): the extract's `stamp`, written LAST        */
    DIF_C_COUNT = 32
};

/* The map: geometry + persistent state (map.py:177-211) + per-map scratch that is all-zero / all -1 between calls. */
typedef struct dif_map {
    int32_t nx, ny, nz;             /* map.py:178 */
    float bound_min[3];             /* map.py:182 */
    float voxel_size;               /* map.py:177 */
    int32_t prune_min_vox_obs;      /* configs/fusion-lr-kt.yaml:33 */
    float ignore_count_th;          /* :34 */
    float encoder_count_th;         /* :35 */
    int64_t capacity;               /* rows of latent_vecs / latent_vecs_pos / voxel_obs_count / dirty / vbm / seg_* */
    int64_t* indexer;               /* [nx*ny*nz] slot or -1                    map.py:201 */
    float* latent_vecs;             /* [capacity][29]                           map.py:204 */
    int64_t* latent_vecs_pos;       /* [capacity] lin id or -1                  map.py:206 */
    float* voxel_obs_count;         /* [capacity]                               map.py:208 */
    uint8_t* dirty;                 /* [capacity] 1 = member of mesh_cache.updated_vec_id (map.py:303-308) */
    uint8_t* voxel_optimized;       /* [capacity]                               map.py:210 (may be NULL if dif_optimize_latents is never called) */
    int32_t* counters;              /* [DIF_C_COUNT] */
    /* scratch, restored to its idle value by every call that touches it */
    int32_t* frame_count;           /* [nx*ny*nz] idle 0  : points of the current frame per voxel (map.py:374) */
    uint32_t* grid_bits;            /* [ceil(nx*ny*nz/32)] idle 0 : candidate / occupied voxel bitmap          */
    int32_t* grid_tot;              /* [1024] idle 0 : set bits of grid_bits per scan block, kept by the kernels that set them */
    int32_t* vbm;                   /* [capacity] idle -1 : vec_id_batch_mapping (map.py:633-635)              */
    int32_t* rec_dir;               /* [capacity][16] idle 0 (words 0,1): the slot's encoder run records of the current integrate */
    int32_t* upd_list;              /* [capacity] slots updated by the current integrate (unique_pinds, map.py:437) */
    int32_t* tri_start;             /* [capacity] mesh-cache log position of the slot's live triangle batch     */
    int32_t* tri_n;                 /* [capacity] idle 0 when the voxel has no cached triangles                 */
    /* Spatial tiling (SURVEY.md section 8e "C5"): this map OWNS the voxels with x index in [own_x_lo, own_x_hi); points whose own
     * voxel lies outside [own_x_lo - halo, own_x_hi + halo) are ignored by integrate, and only owned voxels are meshed.
     * 0, nx, 0 = the whole grid (single-map behaviour). */
    int32_t own_x_lo, own_x_hi, halo;
    /* Optional int32[ceil(capacity / 256)], idle 0: number of set `dirty` flags per block of 256 slots.  dif_integrate* keeps it while it sets flags and dif_extract then compacts the dirty set with ONE launch (and
     * zeroes it again).  A caller that sets flags any other way (dif_merge_*, dif_optimize_latents, its own writes) recomputes the array
     * from the flags afterwards — or passes NULL here, and dif_extract counts the flags itself. */
    int32_t* dirty_tot;
    /* Optional (spatial tiling): int32[2][halo_list_cap], the slots of OWNED voxels in the left boundary layers [own_x_lo, own_x_lo + halo)
     * (row 0) and in the right ones [own_x_hi - halo, own_x_hi) (row 1) that dif_integrate* allocated or fused since the last halo export, in
     * arrival order (a slot may appear more than once); counts in counters[DIF_C_HALO_L / DIF_C_HALO_R].  dif_export_halo_delta turns them into
     * bounded halo messages; NULL: only the whole-layer export (dif_export_halo) is available. */
    int32_t* halo_list;
    int32_t halo_list_cap;
    /* Optional: one dif_pending_export_t in device memory, all-zero when idle.  With it (and dif_extract_buffers_t.defer_export) a
     * streaming caller's copy of an extract's new triangles is not written by that extract but by the FIRST kernel of the next
     * dif_integrate_frame on this map (a few extra workgroups beside the point pass): the PCIe transfer overlaps the next frame
     * instead of lengthening this one.  dif_export_pending does the same copy on its own (last frame of a stream, before a log compaction). */
    void* pending_export;
    /* Optional: a second bitmap + block totals (shapes of grid_bits / grid_tot, idle 0) for the ALLOCATION scan of dif_integrate*, so that it
     * shares nothing with the extract's neighbourhood marker (which keeps grid_bits / grid_tot).  NULL: grid_bits / grid_tot serve both, and a
     * frame's integrate may then not run beside the previous frame's extract. */
    uint32_t* alloc_bits;
    int32_t* alloc_tot;
    /* Two hardware queues for ONE stream of frames (DESIGN.md section 3 "two queues"): with `frame_seq` > 0 and `sync_words` set (`fuse_stream`
     * may be the null stream),
     *   dif_integrate_frame(s)(..., stream A) waits — on the device: hipStreamWaitValue32 — until sync_words[DIF_SYNC_FUSED] >= frame_seq - 1, runs
     *     the frame's FRONT END on A (unproject ... encoder), publishes sync_words[DIF_SYNC_FRONT_DONE] = frame_seq behind it, and enqueues the
     *     fusion kernel on `fuse_stream` (= stream B, the extracts' stream) behind a wait for that word;
     *   dif_extract / dif_extract_streams on stream B publishes sync_words[DIF_SYNC_FUSED] = frame_seq from its first kernel (it starts when the
     *     fusion kernel in front of it on B has completed).
     * So stream B carries fuse(n), extract(n), fuse(n+1), ... back to back with ONE satisfied wait per frame in between, and frame n+1's front
     * end runs on A beside frame n's extract.  What makes that legal: slots that frame n+1 allocates are invisible to frame n's extract
     * (observation count 0 => not confident, batch row -1 => missing: mc_interp_kernel.cu:17-24, map.py:628-631), the allocation bitmap is
     * `alloc_bits`, the dirty-flag block totals are kept by the fusion kernel (behind the extract) instead of the encoder, and the counters of
     * frame n's integrate that its extract hands to the caller are the copies the fusion kernel left in counters[DIF_C_SHADOW ..].  Requires
     * alloc_bits, dirty_tot, frame_counters, capacity > 4096 (a multiple of 256), no deferred export (pending_export idle), an untiled map.  On
     * streams that share a hardware queue (dif_queues_independent) the packets serialise in enqueue order and every wait's producer was enqueued
     * first — correct, nothing gained.  The caller advances
     * frame_seq by one per frame, uses the same value for the frame's integrate and extract, and starts from words that hold frame_seq - 1.
     * frame_seq = 0: off. */
    uint32_t* sync_words;           /* [DIF_SYNC_WORDS] device memory */
    int32_t frame_seq;
    void* fuse_stream;              /* hipStream_t of the extracts: where an overlapped frame's fusion kernel goes */
    /* Overlapped frames: int32[DIF_FC_COUNT] of THIS frame (the caller alternates between two blocks): the fusion kernel leaves the integrate's
     * counters here ([DIF_FC_SHADOW ..]: N_OCCUPIED, ALLOC_NEW, M, C, ITEMS) — what the frame's counter snapshot reports while the next frame's
     * front end already rewrites the live words.  (K, B, VH are written on the extracts' stream only: the live words are the frame's.) */
    int32_t* frame_counters;
} dif_map_t;

/* sync_words (each on a 128-byte line of its own): frame n's fusion kernel has completed (written by the first kernel of its extract); frame n's front
 * end has completed */
enum { DIF_SYNC_FUSED = 0, DIF_SYNC_FRONT_DONE = 32, DIF_SYNC_WORDS = 64 };
enum { DIF_FC_SHADOW = 4, DIF_FC_COUNT = 32 };

/* What a deferred export still has to copy: log rows [kept, kept + n) -> the caller's arrays (see dif_map_t.pending_export). */
typedef struct dif_pending_export {
    int32_t pending, kept, n, seq;          /* seq: the `stamp` of the extract that left this export */
    const float* log_tri; const int64_t* log_id; const float* log_std;
    float* out_tri; int64_t* out_id; float* out_std;
    int32_t* notify;                        /* optional (pinned host memory): receives `seq` once the copy is complete (dif_extract_buffers_t.export_notify) */
} dif_pending_export_t;

/* Network weights packed for the MFMA kernels by di_fusion_amd/network/packing.py (layout documented there). */
typedef struct dif_weights {
    const float* enc_packed;        /* encoder: BN-folded conv weights, per-lane float4 order */
    int64_t enc_packed_floats;
    const float* dec_packed;        /* decoder: weight-norm-folded linear weights */
    int64_t dec_packed_floats;
    const float* dec_bwd_packed;    /* decoder, transposed layers for d sdf / d xyz (may be NULL if gradients are never asked) */
    int64_t dec_bwd_packed_floats;
    const float* dec_fold_packed;   /* decoder, latent columns of lin0 / lin3 for the per-voxel constant folding (packing.py:pack_decoder_fold);
                                     * NULL: every sample row carries the latent through the MFMAs */
    int64_t dec_fold_packed_floats;
    const void* dec_x6_packed;      /* decoder sliced into bf16 triples (packing.py:pack_decoder_x6), or NULL.  When set (together with
                                     * dec_fold_packed) the extract decode tiles run on the bf16 matrix pipe: every fp32 product as six
                                     * exact bf16 slice products, fp32 accumulation — same rounding class as the f32 MFMA, 6/16 of its time */
    int64_t dec_x6_packed_bytes;
    const void* enc_x6_packed;      /* encoder sliced the same way (packing.py:pack_encoder_x6), or NULL: integrate's encoder tiles and
                                     * dif_encode_rows run on the bf16 matrix pipe when set */
    int64_t enc_x6_packed_bytes;
    const void* dec_x6u_packed;     /* with dec_x6_packed: slices of lin0 and of lin3's skip block (packing.py:pack_decoder_x6u) for the tiles that
                                     * carry the whole 32-column input (dif_decode_rows, dif_query_sdf without gradient, non-fast extract) */
    int64_t dec_x6u_packed_bytes;
    const void* dec_x6b_packed;     /* with the two above: slices of the transposed layers (packing.py:pack_decoder_x6_backward): dif_query_sdf with
                                     * gradient on the bf16 matrix pipe */
    int64_t dec_x6b_packed_bytes;
} dif_weights_t;

int dif_version(void);
/* "<hash>:<hipcc version>": hash of the sources this library was built from (everything under csrc/, this header, the compiler flags;
 * di_fusion_amd/_build.py computes it and passes it as -DDIF_BUILD_ID) — the loader compares it with the tree and rebuilds on a mismatch
 * instead of trusting file times — and the version of the compiler that built it. */
const char* dif_build_id(void);

/* ---- a1/a2: depth -> points (ext/imgproc/imgproc.cu:5-44, utils/motion_util.py:322-327) -------------------- */
/* pc[v][u] = ((u-cx)/fx*d, (v-cy)/fy*d, d); NaN depth -> (NaN,NaN,NaN).  depth (H,W) f32 -> pc (H,W,3) f32.     */
int dif_unproject(const float* depth, float* pc, int32_t H, int32_t W, float fx, float fy, float cx, float cy,
                  void* stream);
/* Fused: unproject, then world = R*p + t and n_world = R*n_cam (R row-major 3x3 [host], t[3] [host]).
 * normal_cam may be NULL.  Outputs (H*W,3); invalid pixels are NaN rows (masked out by integrate).              */
int dif_unproject_transform(const float* depth, const float* normal_cam, float* xyz_world, float* normal_world,
                            int32_t H, int32_t W, float fx, float fy, float cx, float cy,
                            const float* R, const float* t, void* stream);
/* Same with the pose in DEVICE memory (12 floats: R row-major, then t), so the call can sit in a captured hipGraph that is
 * replayed every frame with a new pose copied into `pose_dev`. */
int dif_unproject_transform_dev(const float* depth, const float* normal_cam, float* xyz_world, float* normal_world,
                                int32_t H, int32_t W, float fx, float fy, float cx, float cy, const float* pose_dev,
                                void* stream);
/* Same, with the whole frame input behind one device-resident descriptor: a captured hipGraph of the per-frame chain is replayed
 * on new inputs after a single 64-byte upload (no staging copies of depth / normals into fixed buffers). */
typedef struct dif_frame {
    const float* depth;        /* (H,W) f32, device */
    const float* normal_cam;   /* (H,W,3) f32, device */
    float pose[12];            /* R row-major (9) then t (3) */
} dif_frame_t;
int dif_unproject_transform_frame(const dif_frame_t* frame_dev, float* xyz_world, float* normal_world, int32_t H, int32_t W,
                                  float fx, float fy, float cx, float cy, void* stream);
/* ext/imgproc/imgproc.cu:98-160: pc (H,W,3) -> normal_weight (H,W,4), w=-1 where invalid. */
int dif_compute_normal_weight(const float* pc, float* normal_weight, int32_t H, int32_t W, void* stream);

/* ---- 8f-2: image-space preprocessing on either side of the path ----------------------------------------------------- */
/* ext/imgproc/imgproc.cu:48-94: 5x5 bilateral depth filter (range sigma from the depth-noise model); the 2-pixel border of
 * depth_out is left untouched, depth < 1e-6 -> 0. */
int dif_filter_depth(const float* depth_in, float* depth_out, int32_t H, int32_t W, void* stream);
/* The three image-space kernels of ext/imgproc as ONE pass (16 x 16 pixel tiles, the depth tile and its apron staged through LDS):
 *   depth_out     = filter_depth(depth) with the 2-pixel border copied from `depth`   (imgproc.cu:48-94; filter == 0: depth_out = depth)
 *   pc            = unproject_depth(depth_out)                                        (imgproc.cu:5-44)
 *   normal_weight = compute_normal_weight(pc)                                         (imgproc.cu:98-160)
 * bit-identical to calling the three entry points one after the other.  Any of the outputs may be NULL.  frame_depth / frame_normal
 * (both or neither): depth_out where a normal exists and NaN elsewhere, and that normal — the two arrays a dif_frame_t points at, so
 * that a stream WITHOUT supplied normals can go straight into dif_integrate_frame. */
int dif_depth_frontend(const float* depth, int32_t H, int32_t W, float fx, float fy, float cx, float cy, int32_t filter, float* depth_out,
                       float* pc, float* normal_weight, float* frame_depth, float* frame_normal, void* stream);
/* system/tracker.py:13-23 point_box_filter: mean point and mean normal per voxel_size box; boxes come out in ascending
 * linear box id (x fastest), out_count[0] (device) = number of boxes.  out_points/out_normals: (N,3) capacity.
 * bits: uint32[(max_cells+31)/32] all-zero on entry and on exit; word_rank: int32[(max_cells+31)/32]; sums: int64[N*8];
 * scratch: int32[4200].  If the box grid of the cloud exceeds max_cells, scratch[4102] is set and nothing is produced. */
int dif_point_box_filter(const float* points, const float* normals, int64_t N, float voxel_size, float* out_points,
                         float* out_normals, int32_t* out_count, uint32_t* bits, int64_t max_cells, int32_t* word_rank,
                         int64_t* sums, int32_t* scratch, void* stream);

/* ---- 8f-3: point-cloud neighbourhood ops of the tracker's pre-processing (system/tracker.py:105-113) --------------------- */
/* The reference builds a FLANN-derived CUDA kd-tree per call (ext/pcproc/cuda_kdtree.cu:644-960) and runs an exact kNN
 * (nearestKernel, :1070; squared L2 of CudaL2::dist, :1152-1155; results ascending by distance).  Here the index is a hashed
 * uniform grid living in the caller's workspace and the search is exact below `radius`: among the k nearest points of a
 * query (itself included, order (d2, index)), every one with d2 < radius^2 is reported exactly; the others as idx -1,
 * dist +inf.  pc is (n, stride) with stride 3 or 4 (the tracker passes xyz0 rows); 1 <= k <= 32.  Points with a non-finite
 * coordinate have no neighbours and are nobody's neighbour. */
int64_t dif_cloud_workspace_bytes(int64_t n);
int dif_knn(const float* pc, int64_t n, int32_t stride, int32_t k, float radius, int32_t* out_idx, float* out_dist, void* workspace,
            int64_t workspace_bytes, void* stream);
/* ext/pcproc/pcproc.cu:98-105,160-186 remove_radius_outlier: out_mask[i] = (squared distance to the nb_points-th nearest
 * point, self included) < radius^2. */
int dif_remove_radius_outlier(const float* pc, int64_t n, int32_t stride, int32_t nb_points, float radius, uint8_t* out_mask,
                              void* workspace, int64_t workspace_bytes, void* stream);
/* ext/pcproc/pcproc.cu:107-158,188-209 estimate_normals: PCA normal over neighbours 1..max_nn-1 of the sorted kNN list while
 * inside `radius` (fewer than 5 -> NaN), eigenvector by pcproc.cu:21-96, oriented towards cam_xyz (host float[3]). */
int dif_estimate_normals(const float* pc, int64_t n, int32_t stride, int32_t max_nn, float radius, const float* cam_xyz,
                         float* out_normals, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- a9: ext/indexing/indexing.cu:89-109 ------------------------------------------------------------------- */
/* sum[idx[i]][:] += values[i][:], count[idx[i]] += 1 (per sample).  sum (C,L) / count (C) must be zeroed by the
 * caller.  Deterministic is not promised here (float atomics, as the reference); the map path does not use it.  */
int dif_groupby_sum(const float* values, const int64_t* indices, int64_t N, int32_t L, float* sum, int32_t* count,
                    int64_t C, void* stream);

/* ---- a3..a10: integrate_keyframe (map.py:340-519, do_optimize=False) --------------------------------------- */
/* Bytes of scratch `ws` needed for N points: the point ids, the compacted list of gathered (voxel, offset, point) rows (at most 8N,
 * map.py:419-435) and the encoder's per-run partial sums (at most one 256-byte record per row; written sparsely). */
int64_t dif_integrate_workspace_bytes(int64_t N);
/* xyz, normal: (N,3) f32.  unq_mask: (N) u8 out (map.py:375; all-valid-points when prune_min_vox_obs<=0).
 * Points with NaN coordinates or outside [bound_min, bound_max) are masked out (the reference indexes out of
 * bounds there, map.py:313; documented divergence). */
int dif_integrate(const dif_map_t* map, const dif_weights_t* w, const float* xyz, const float* normal, int64_t N,
                  uint8_t* unq_mask, void* ws, int64_t ws_bytes, void* stream);

/* a1 + a2 + a3..a10 for a streaming caller: integrate straight from a depth frame behind a device-readable descriptor (see
 * dif_unproject_transform_frame).  The first kernel back-projects, transforms AND counts points per voxel.  xyz_world / normal_world
 * ((H*W,3) each) are optional outputs (both or neither): with them the later stages read the points there; without, the few points those
 * stages need (the ~3 % that pass the focus test, the gathered encoder rows) are recomputed from the depth pixel with the same operations —
 * nothing H*W*24 bytes large is written per frame.  Same results either way, and the same as dif_unproject_transform_frame followed by
 * dif_integrate. */
int dif_integrate_frame(const dif_map_t* map, const dif_weights_t* w, const dif_frame_t* frame_dev, int32_t H, int32_t W, float fx, float fy,
                        float cx, float cy, float* xyz_world, float* normal_world, uint8_t* unq_mask, void* ws, int64_t ws_bytes, void* stream);

/* ---- 8f-4: integrate_keyframe(do_optimize=True), stage 3 (map.py:459-513, OptimizeProcess.do_optimize :80-113, write-back :321-335) ----
 * To be called right after dif_integrate on the SAME points and with the unq_mask it produced.  Voxels with observation count >=
 * encoder_count_th that were never optimised (and lin id > 0, as there) get `n_iters` Adam steps (lr, torch defaults otherwise) on
 *   sum_rows -log N(clamp(gt); clamp(sdf), std) / n_rows  [+ code_reg_lambda * sum_voxels |z| / n_rows  when code_reg_lambda > 0]
 * over the perturbed surface samples gathered around them (8 offsets; row k is displaced along its normal by 0.05 * noise[k], rows in
 * the reference's gathering order: offset-major, then point order), then their latents are written back and they are marked optimised
 * and dirty.  noise: device, at least 8N floats of N(0,1) samples (the reference draws them from torch.randn).  loss_out: optional
 * device float[64], the likelihood part of the loss before each of the first 64 steps.  Counts -> DIF_C_OPT_ROWS / DIF_C_OPT_VOXELS.
 * Nothing synchronises. */
int64_t dif_optimize_workspace_bytes(int64_t N, int64_t capacity);
int dif_optimize_latents(const dif_map_t* map, const dif_weights_t* w, const float* xyz, const float* normal, int64_t N,
                         const uint8_t* unq_mask, const float* noise, int32_t n_iters, float lr, float code_reg_lambda, float* loss_out,
                         void* ws, int64_t ws_bytes, void* stream);

/* ---- a11..a16: extract_mesh (map.py:581-723) ---------------------------------------------------------------- */
typedef struct dif_extract_buffers {
    int64_t max_voxels;             /* rows available in the per-voxel buffers below                      */
    int64_t* valid_blocks;          /* [max_voxels]  lin ids of dirty voxels, ascending slot order (map.py:627) */
    int32_t* occ_slot;              /* [max_voxels]  slots of the B decoded voxels, ascending lin id      */
    float* low_sdf;                 /* [max_voxels][l^3]   l = resolution                                 */
    float* low_std;
    float* cube_sdf;                /* [max_voxels][R^3]   R = 2*resolution, NEGATED sdf (map.py:687)     */
    float* cube_std;
    int32_t* refine_list;           /* [max_voxels*R^3]   b*R^3 + sb of samples to re-decode (map.py:667) */
    int32_t* tri_count;             /* [max_voxels] triangles per dirty voxel                             */
    int32_t* tri_offset;            /* [max_voxels] exclusive prefix of tri_count                         */
    int32_t* block_tmp;             /* [4096] scan scratch                                                */
    int64_t max_triangles;          /* map.py:581 max_n_triangles (per call)                              */
    /* Device-resident mesh cache (the reference keeps it in host numpy arrays and rebuilds them on every extract, map.py:703-714).
     * Here it is an append-only LOG: an extract appends its new triangles (canonical order) and kills the previous batch of
     * every voxel that produced >= 1 new triangle (`map->tri_start/tri_n` locate it) -- O(new + replaced) per extract instead
     * of O(whole mesh).  The live entries, in log order, ARE the reference's `mesh_cache.vertices / vertices_flatten_id /
     * vertices_std` arrays (kept old triangles in their order, then the new ones): `dif_mesh_cache_compact` materialises them.
     * The new triangles of a call are log[DIF_C_CACHE_KEPT : DIF_C_CACHE_T]. */
    int64_t cache_capacity;         /* log capacity in triangles                                          */
    float* cache_tri;               /* [cache_capacity][3][3]                                             */
    int64_t* cache_id;              /* [cache_capacity]                                                   */
    float* cache_std;               /* [cache_capacity][3]                                                */
    uint8_t* cache_alive;           /* [cache_capacity]                                                   */
    int32_t* counters_out;          /* optional [DIF_C_COUNT]: the map's counters as of the end of this extract, written by its last
                                     * kernel; may be device-mapped pinned HOST memory (no copy kernel, readable after a stream sync) */
    float* out_tri;                 /* optional: this call's NEW triangles (log[DIF_C_CACHE_KEPT : DIF_C_CACHE_T], the first out_capacity   */
    int64_t* out_id;                /* of them) copied out by the same last kernel — (n,3,3) f32, (n) i64, (n,3) f32; again, pinned host   */
    float* out_std;                 /* memory is fine: a streaming caller gets each frame's mesh update without a transfer of its own      */
    int64_t out_capacity;
    int32_t* chunk_sum;             /* optional [(max_voxels + 255) / 256 + (max_voxels + 65535) / 65536], idle 0: triangle counts per 256 and per
                                     * 65,536 dirty voxels; with it (and max_voxels <= 2^24) marching cubes runs as two launches (count, emit)
                                     * instead of count, scan, emit */
    float* fold_table;              /* optional [max_voxels][256]: per-voxel decoder constants handed from the lattice decode to the refine
                                     * decode (used when dif_weights_t.dec_fold_packed is set) */
    uint32_t* mc_status;            /* optional [(max_voxels + 3) / 4 + 256], idle 0: with it (and chunk_sum, resolution <= 4) marching cubes is ONE launch:
                                     * a wave counts its voxel's triangles, learns its output offset by a decoupled look-back over groups of
                                     * four voxels and emits straight away (same canonical order).  A call with more groups than the launch has
                                     * workgroups (thousands of dirty voxels) hands the groups out through eight ticket words (one per XCD, 32
                                     * words apart) at the end of this array (dif_test_mc_grid_cap caps that launch, so that small maps take
                                     * the ticket path in tests) */
    int32_t defer_export;           /* != 0 (with out_* and dif_map_t.pending_export): do not copy the new triangles to out_* now, leave a
                                     * dif_pending_export_t for the next dif_integrate_frame / dif_export_pending */
    /* Completion without stream events (an event record costs the queue ~5 us between two kernels; a frame has two):
     *   stamp          — written into counters_out[DIF_C_STAMP] by the extract's last kernel AFTER the other words (system-scope fence in
     *                    between): a caller that polls that word of its pinned snapshot for the stamp it passed has the whole snapshot.  Valid
     *                    as a completion signal for out_* too when the triangles were written by the one-pass marching cubes or deferred —
     *                    not when the last kernel itself copies them (two-pass marching cubes without defer_export);
     *   export_notify  — with defer_export: int32 in pinned host memory that receives `stamp` from the last kernel of the
     *                    dif_integrate_frame(s) that carried the deferred copy out (its point kernels copy, its fusion kernel notifies). */
    int32_t stamp;
    int32_t* export_notify;
} dif_extract_buffers_t;

/* resolution r (map.py:581 voxel_resolution; lattice R=2r), fast!=0: two-level decode (low lattice l=r, trilinear x2,
 * exact re-decode where |sdf|<0.05; map.py:655-682) else all R^3 samples decoded; max_std (mc_interp_kernel.cu:304),
 * no_cache!=0 re-meshes every allocated voxel (map.py:614-616), scale_vertices!=0 writes v*voxel_size+bound_min
 * (map.py:698) instead of voxel units.  Triangles come out in canonical order (dirty voxel, cell, table order). */
int dif_extract(const dif_map_t* map, const dif_weights_t* w, const dif_extract_buffers_t* buf, int32_t resolution,
                int32_t fast, float max_std, int32_t no_cache, int32_t scale_vertices, void* stream);

/* Perform the copy a deferred export left pending (dif_map_t.pending_export), if any, and mark it done. */
int dif_export_pending(const dif_map_t* map, void* stream);

/* ---- several independent subsequences per launch (SURVEY.md section 8e "C4", here INSIDE one GPU) ------------------------------------
 * A steady-state frame of ONE stream is 1-2 MLP tiles per SIMD and nine launches that sit on their latency floors.  These two entry points
 * run the frame of S <= DIF_MAX_STREAMS independent streams (S private maps over grids of the same shape, S frames of the same size)
 * through the SAME twelve launches: the point / scan / fusion / marching-cubes kernels get a second grid dimension (blockIdx.y = stream),
 * the persistent encoder / decoder kernels walk the streams' tiles as one concatenated range (weights staged once per workgroup).
 * Every stream's results are bit-identical to dif_integrate_frame + dif_extract on that stream alone.
 * Requirements (else DIF_EINVAL): the maps have the same nx, ny, nz and capacity, are not spatially tiled, carry dirty_tot / grid_tot /
 * pending_export like a streaming map; the extract buffers have the same max_voxels, chunk_sum / mc_status / fold_table set; the weights
 * carry the bf16-sliced blobs (the default pipe); resolution <= 4, fast two-level decode. */
typedef struct dif_stream_frame {
    const dif_map_t* map;                   /* [host] */
    const dif_frame_t* frame_dev;           /* as dif_integrate_frame */
    float* xyz_world; float* normal_world;  /* (H*W,3) each, out; or both NULL (see dif_integrate_frame) */
    uint8_t* unq_mask;                      /* (H*W) out */
    void* ws; int64_t ws_bytes;             /* dif_integrate_workspace_bytes(H*W) */
    const dif_extract_buffers_t* buf;       /* [host] as dif_extract (dif_extract_streams only; may be NULL for dif_integrate_frames) */
} dif_stream_frame_t;
int dif_integrate_frames(const dif_stream_frame_t* streams /* [host][S] */, int32_t S, const dif_weights_t* w, int32_t H, int32_t W, float fx,
                         float fy, float cx, float cy, void* stream);
int dif_extract_streams(const dif_stream_frame_t* streams /* [host][S] */, int32_t S, const dif_weights_t* w, int32_t resolution, float max_std,
                        int32_t scale_vertices, void* stream);

/* Log entries [lo, lo+n) -> (out_tri, out_id, out_std) in one launch.  The destinations may be device-mapped pinned HOST memory: a
 * streaming caller ships each call's new triangles (lo = DIF_C_CACHE_KEPT, n = DIF_C_CACHE_T - lo) without copy-engine transfers. */
int dif_mesh_cache_export(const dif_extract_buffers_t* buf, int64_t lo, int64_t n, float* out_tri, int64_t* out_id, float* out_std,
                          void* stream);
/* The same rows by the COPY ENGINE (three hipMemcpyAsync device -> pinned host on `stream`, no kernel): for a caller that knows the frame
 * is complete — it has seen the frame's stamp (dif_extract_buffers_t.stamp), or `stream` is ordered behind the extract — and wants the
 * transfer beside the next frame's kernels instead of inside them.
 * Replaces the host copy of map.py:703-714 (`.cpu().numpy()` of the three marching-cubes outputs). */
int dif_mesh_cache_export_dma(const dif_extract_buffers_t* buf, int64_t lo, int64_t n, float* out_tri, int64_t* out_id, float* out_std,
                              void* stream);
/* Materialise the live cache entries in log order into (out_tri, out_id, out_std); count -> counters[DIF_C_CACHE_LIVE].
 * scratch: int32 [4096]. */
int dif_mesh_cache_compact(const dif_map_t* map, const dif_extract_buffers_t* buf, float* out_tri, int64_t* out_id, float* out_std,
                           int64_t out_capacity, int32_t* scratch, void* stream);
/* After the caller has swapped a compacted copy in as the new log (n = its length): rebuild tri_start / tri_n, mark everything
 * alive, set CACHE_T = CACHE_KEPT = n, CACHE_DEAD = 0. */
int dif_mesh_cache_reindex(const dif_map_t* map, const dif_extract_buffers_t* buf, int64_t n, void* stream);

/* Flat marching cubes = ext/marching_cubes mc.cpp:3-16.  indexer (nx,ny,nz) i64, valid_blocks (K) i64,
 * vec_batch_mapping (V) i32, cube_sdf/std (B,R,R,R) f32.  counters[DIF_C_T] receives the triangle count.      */
int dif_marching_cubes(const int64_t* indexer, int32_t nx, int32_t ny, int32_t nz, const int64_t* valid_blocks,
                       int64_t K, const int32_t* vec_batch_mapping, int64_t V, const float* cube_sdf,
                       const float* cube_std, int32_t R, float max_std, int64_t max_triangles, float* triangles,
                       int64_t* triangle_flatten_id, float* triangle_std, int32_t* tri_count, int32_t* tri_offset,
                       int32_t* block_tmp, int32_t* counters, void* stream);

/* ---- a13/a17: decoder on explicit rows (network/utility.py:61-126, map.py:559-579) -------------------------- */
/* rows (n,32) = [latent 29 | xyz 3] -> sdf (n), std (n). */
int dif_decode_rows(const dif_weights_t* w, const float* rows, int64_t n, float* sdf, float* std_out, void* stream);
/* encoder on explicit rows (network/di_encoder.py:26-30): rows (n,6) -> out (n,29).  Test / ext entry point.   */
int dif_encode_rows(const dif_weights_t* w, const float* rows, int64_t n, float* out, void* stream);
/* get_sdf: xyz (N,3) -> mask (N) u8, and for the M valid points IN ORDER: sdf (M), std (M), grad (M,3) = d sdf / d xyz
 * in world units (NULL to skip; reference tracker.py:186-192 obtains it by autograd), sel (M) = index of the point.  M -> counters[DIF_C_QUERY_M].
 * scratch: int32 [N + 4096]; on return scratch[4096 + i] = row of point i among the valid ones or -1 (the inverse of sel). */
int dif_query_sdf(const dif_map_t* map, const dif_weights_t* w, const float* xyz, int64_t N, uint8_t* mask,
                  int32_t* sel, float* sdf, float* std_out, float* grad, int32_t* scratch, void* stream);
/* The same in two steps, for a caller that has to hand back M-row tensors (map.py:559-579 does) without waiting for the decoder:
 * dif_query_select = validity mask + ordered compaction (mask, sel, M -> counters[DIF_C_QUERY_M]); count_out (optional, device-mapped pinned
 * HOST memory allowed): [0] = M, [1] = seq, written by the compaction's last workgroup — the host can slice its outputs as soon as that
 * small kernel is done.  dif_query_decode = the decoder over those M rows (grad may be NULL), reading M on the device. */
int dif_query_select(const dif_map_t* map, const float* xyz, int64_t N, uint8_t* mask, int32_t* sel, int32_t* scratch, int32_t* count_out,
                     int32_t seq, void* stream);
int dif_query_decode(const dif_map_t* map, const dif_weights_t* w, const float* xyz, int64_t N, const int32_t* sel, float* sdf,
                     float* std_out, float* grad, void* stream);
/* Backward of get_sdf for the caller's autograd (the reference differentiates through the decoder, tracker.py:186-192):
 * out[sel[m]][:] = grad[m][:] * g_sdf[m] for m < M; out (N,3) is zeroed by the caller. */
int dif_query_grad_scatter(const float* grad, const float* g_sdf, const int32_t* sel, int64_t M, float* out, void* stream);
/* The same in one launch WITHOUT the zero fill: every row of out (N,3) is written, through the inverse of `sel` that dif_query_select leaves in
 * its scratch (scratch[4096 + i] = row of point i among the valid ones, -1 = invalid).  `scratch`: the array that call was given, untouched since. */
int dif_query_grad_gather(const float* grad, const float* g_sdf, const int32_t* scratch, int64_t N, float* out, void* stream);

/* ---- f1: the SDF term of the tracker's Gauss-Newton step in one call (tracker.py:174-218, SDFTracker.compute_sdf_Hg) ---------------------------
 * obs_xyz (N,3): the frame's points in CAMERA space (tracker.py:220 obs_xyz).  Per call:
 *   cur = (last_pose . cur_delta_pose) @ obs (tracker.py:181; float32 [R | t] rows as Isometry.torch_matrices gives them, motion_util.py:322-327),
 *   get_sdf(cur) with d sdf / d cur (map.py:559-579, tracker.py:184-192), residual s = sdf / std,
 *   J = [d @ last_R^T | (cur_delta_pose @ obs) x (d @ last_R^T)] with d = grad(sdf) / std (tracker.py:193-199),
 *   robust weights (tracker.py:58-72: 1 = "huber", 2 = "tukey", 0 = none), and the sums scaled by 1 / M (tracker.py:209-218).
 * out (device, 44 doubles): H row-major [0,36), g [36,42), sum_error [42], M [43].  M = 0 -> all zero (the reference divides by zero there).
 * out_host (optional, pinned host memory the device can write, 45 x 8 bytes): the same 44 doubles, then `seq` as int64 behind a system-scope
 * fence — a host that polls word 44 for its sequence number has the numbers without a copy or a stream synchronisation.
 * no_grad != 0: only sum_error and M (the loop's last evaluation, tracker.py:239), through the value-only decoder.
 * Two launches: the decoder kernel over all N points (pose, validity test and latent look-up in its row fetch; invalid points are marked, not
 * compacted) and the reduction.  ws: dif_sdf_hg_workspace_bytes(N) bytes of device memory, 256-byte aligned, private to the call's stream while it runs.
 * The sums are accumulated in double in a fixed order: the same inputs give the same bits. */
typedef struct dif_sdf_hg_t {
    float T_cur[12];        /* last_pose . cur_delta_pose: rows of [R | t]                         */
    float T_delta[12];      /* cur_delta_pose                                                      */
    float last_Rt[9];       /* last_pose.q.rotation_matrix.T, row-major (tracker.py:196 `Lt`)      */
    int32_t robust_kernel;  /* 0 none, 1 huber, 2 tukey                                            */
    float robust_k;
    int32_t no_grad;
} dif_sdf_hg_t;
int64_t dif_sdf_hg_workspace_bytes(int64_t N);
int dif_sdf_hg(const dif_map_t* map, const dif_weights_t* w, const float* obs_xyz, int64_t N, const dif_sdf_hg_t* args, void* ws, int64_t ws_bytes,
               double* out, double* out_host, int64_t seq, void* stream);

/* ---- multi-GPU map merge (no reference counterpart) ----------------------------------- */
/* Pack the allocated voxels whose x index lies in [x_lo, x_hi) as 32-word records, in slot order:
 *   lin int32 | flags int32 (bit 0 = dirty) | w f32 | payload f32[29],  payload = w*z (raw == 0: additive merge) or z itself (raw != 0: exact copy).
 * The number written goes to counters[DIF_C_EXPORT_N].  scratch: int32 [4096]. */
int dif_export_records(const dif_map_t* map, int32_t* records, int64_t max_records, int32_t x_lo, int32_t x_hi, int32_t raw,
                       int32_t* scratch, void* stream);
/* Fold `n` records WITH DISTINCT lin ids into the map; unseen voxels are allocated in ascending lin order first.
 *   assign == 0 (records exported with raw == 0):  w += w_r ; z = (z*w + wz_r) / (w + w_r) ; dirty if w_r > 0   (map merge, C4)
 *   assign != 0 (records exported with raw != 0):  w = w_r ; z = z_r ; dirty = flags & 1                         (halo copy, C5)
 * scratch: int32 [4096]. */
int dif_merge_records(const dif_map_t* map, const int32_t* records, int64_t n, int32_t assign, int32_t* scratch, void* stream);

/* Halo messages for the spatially tiled mode (C5): fixed-capacity buffers whose LENGTH travels inside them, so a frame's exchange needs
 * no host round trip (no counter read-back between export, send/recv and merge) and always moves the same number of bytes.
 *   message = int32 [1 + max_records][32]: row 0 = header (word 0 = number of records that follow), rows 1.. = raw records
 *   (lin | flags | w | z[29], as dif_export_records with raw != 0) of the allocated voxels with x index in [x_lo, x_hi), slot order.
 * dif_merge_halo folds a received message with assign semantics (w = w_r, z = z_r, dirty = flags & 1), reading the count on the
 * device (clamped to max_records).  scratch: int32 [4096]. */
int dif_export_halo(const dif_map_t* map, int32_t* message, int64_t max_records, int32_t x_lo, int32_t x_hi, int32_t* scratch, void* stream);
int dif_merge_halo(const dif_map_t* map, const int32_t* message, int64_t max_records, int32_t* scratch, void* stream);
/* Bounded DELTA messages: only the boundary voxels that dif_integrate* allocated or fused since the last halo export (map->halo_list), for
 * both neighbours in ONE launch.  A receiver whose halo copy was exact before the frame is exact again after merging the delta, because
 * integrate never writes a voxel it does not own (its gather drops those targets).  msg_left / msg_right (NULL: no delta message for that
 * side — its change list is left alone): int32 [1 + max_records][32]; header word 0 = records that follow, word 1 = records that were
 * pending (> word 0: the message or the change list overflowed, counters[DIF_C_OVERFLOW] = 8 and the receiver's halo is stale from here on),
 * word 2 = 1.  The change list of every side that got a message is emptied.
 * dif_export_halo writes header word 0 only; dif_halo_lists_reset completes the headers of such whole-layer messages (word 1 = entries
 * that were pending in that side's list, i.e. the size a delta message would have had — what the host's choice between the two message
 * kinds looks at —, word 2 = 0; NULL: side not touched) and empties those sides' lists, which the whole-layer export supersedes.
 * note (optional, all three calls; may be device-mapped pinned HOST memory): int32[8] = header words 0..3 of the left / first message, then
 * of the right / second one, as written (exports) or as received (merge): the host reads them a frame or two later, without a copy. */
int dif_export_halo_delta(const dif_map_t* map, int32_t* msg_left, int32_t* msg_right, int64_t max_records, int32_t* note, void* stream);
int dif_halo_lists_reset(const dif_map_t* map, int32_t* header_left, int32_t* header_right, int32_t* note, void* stream);
/* dif_merge_halo for the messages of both neighbours in one pass (three launches instead of six); either may be NULL. */
int dif_merge_halo2(const dif_map_t* map, const int32_t* msg_a, int64_t max_a, const int32_t* msg_b, int64_t max_b, int32_t* scratch,
                    int32_t* note, void* stream);

/* ---- per-kernel timing for bench.py's roofline leg ---------------------------------------------------------- */
/* When enabled, a hipEvent pair is recorded around each launch of the named kernels ON THE STREAM THEY RUN ON. */
enum { DIF_PROF_ENCODE = 0, DIF_PROF_DECODE_LATTICE = 1, DIF_PROF_DECODE_POINTS = 2, DIF_PROF_MC_COUNT = 3, DIF_PROF_MC_EMIT = 4,
       DIF_PROF_HALO_EXPORT = 5 /* the launches of one dif_export_halo* call */, DIF_PROF_HALO_MERGE = 6 /* of one dif_merge_halo* call */,
       DIF_PROF_COUNT = 8 };
int dif_profile_enable(int32_t on);
/* Sum of elapsed milliseconds and number of launches per kernel since the last reset; synchronises on the events. */
int dif_profile_read(double* ms /* [DIF_PROF_COUNT], host */, int64_t* launches /* [DIF_PROF_COUNT], host */, int32_t reset);
/* The same per launch, in launch order: which[i] (DIF_PROF_*), ms[i]; returns the number of records written (<= capacity) or a
 * negative DIF_E* code.  Thread-safe with respect to concurrent launches (the record list is mutex-guarded). */
int64_t dif_profile_dump(int32_t* which /* host */, float* ms /* host */, int64_t capacity, int32_t reset);

/* Copy the counters to the host; the only synchronising call (hipStreamSynchronize on `stream`). */
int dif_read_counters(const dif_map_t* map, int32_t* host_out /* [DIF_C_COUNT], host */, void* stream);

/* 1 if work on streams `a` and `b` really runs concurrently — they sit on different hardware queues —, 0 if not, negative on error: a kernel on `a`
 * waits (bounded: ~20 ms) for a word that a kernel enqueued LATER on `b` writes.  HIP shares a hardware queue between streams once more than
 * GPU_MAX_HW_QUEUES (default 4) are alive; a device-side wait (dif_map_t.frame_seq) between two streams that share one would never end.
 * Synchronises both streams. */
int dif_queues_independent(void* stream_a, void* stream_b);

/* The same rows as dif_mesh_cache_export_dma, by the SDMA engines directly (hsa_amd_memory_async_copy: no copy kernel on any queue, nothing beside
 * the frame's kernels): three copies under one completion signal, issued from the host NOW — the caller has seen the frame complete (its stamp) —
 * and waited for (the call returns when the rows are in `out_*`, pinned host memory).  DIF_ELAUNCH when the HSA runtime of the process cannot be
 * reached or refuses the copy (the caller then uses dif_mesh_cache_export_dma). */
int dif_mesh_cache_export_sdma(const dif_extract_buffers_t* buf, int64_t lo, int64_t n, float* out_tri, int64_t* out_id, float* out_std);

/* TEST HOOK, not part of the reference's interface: caps the launch of the one-pass marching cubes (dif_extract / dif_extract_streams) at n workgroups
 * (n <= 0: no cap, the default), so that a small map takes the ticket path that otherwise only a map with thousands of dirty voxels takes.  Process-wide;
 * returns the previous cap.  (Until round 5 this was the environment variable DIF_MC_GRID, read at every call.) */
int dif_test_mc_grid_cap(int32_t n);
/* TEST HOOK: how dif_mesh_cache_export_sdma behaves from now on (process-wide; returns the previous mode): 0 = default; 1 = leave the engine choice to
 * the runtime (what happens under a HIP runtime other than the validated one); 2 = fail with DIF_ELAUNCH (what happens when the process's HSA runtime
 * cannot be reached: the caller falls back to dif_mesh_cache_export_dma); 3 = count every export as slow (the engines are timed again after four). */
int dif_test_sdma_mode(int32_t mode);
/* TEST HOOK: a litmus run of the fence-free hand-overs the product kernels use (write-through stores + s_waitcnt vmcnt(0) + word on the producer,
 * sc1 loads on the consumer; csrc/kernels_litmus.hip.h).  mode 0: `groups` workgroups meet `iters` times through a counter every one polls, each then
 * checks another workgroup's 29-double record (a launch-wide meeting: the pattern of profiles/r06_experiments.md 1); mode 1: the last arriver of a ticket checks all records
 * (k_sdf_hg_reduce's); mode 2: `groups` workgroups hand 44 doubles + a sequence word to the CPU through pinned memory `iters` times
 * (k_sdf_hg_reduce's / k_extract_finish's hand-back).  flags bit 0: on a stream confined to every other CU; bit 1: beside a kernel that streams
 * through 1 GB.  out (host, int64[4]): stale values seen, hand-overs checked, time-outs, microseconds.  Allocates and frees what it needs;
 * synchronises. */
int dif_test_handoff(int32_t mode, int32_t groups, int32_t iters, int32_t flags, int64_t* out);
/* out (host, int32[8]): HSA runtime reached (0/1), hipRuntimeGetVersion of the process, the version the engine selection was validated with,
 * engine calibrations so far, exports so far, the three engines in use (-1: the runtime's choice). */
int dif_sdma_info(int32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* DIFUSION_H */
