// integrate_keyframe kernels: a3-a10 voxel ids, prune, allocate, gather, encoder (MFMA), fusion  (part of libdifusion; included by difusion.hip inside its anonymous namespace)
#pragma once

// =================================================================================================================
// a3..a6 : voxel ids, prune, allocate   (map.py:366-387)
// =================================================================================================================
// K1: per-point voxel id + per-voxel point count of this frame.
// Per-voxel point counts of the frame (map.py:374) with wave-run aggregation: pixels of a row that fall into the same voxel are
// neighbours in the wave, so one atomic per run of equal ids.  Also zeroes the per-call counters.
__device__ __forceinline__ void voxel_count_point(const Geo& g, bool in_range, float x, float y, float z, int64_t i, int* __restrict__ pt_lin,
                                                  int* __restrict__ frame_count, int* __restrict__ counters, int px_lo, int px_hi) {
    const int lane = lane_id();
    if (i < 4) counters[DIF_C_ALLOC_NEW + i] = 0;                   // ALLOC_NEW, M, C, ITEMS of this call
    int lin = -2;                                                    // -2: beyond N, -1: invalid point
    if (in_range) {
        float xn, yn, zn; int ix, iy, iz;
        bool ok = voxel_of(g, x, y, z, xn, yn, zn, ix, iy, iz);
        ok = ok && ix >= px_lo && ix < px_hi;                         // spatial tiling: own slab + halo only
        lin = ok ? linearize(g, ix, iy, iz) : -1;
        pt_lin[i] = lin;
    }
    int prev = __shfl_up(lin, 1);
    bool head = (lane == 0) || (prev != lin);
    unsigned long long heads = __ballot(head);
    if (head && lin >= 0) {
        unsigned long long above = (lane == 63) ? 0ull : (heads >> (lane + 1));
        int run = above ? __ffsll((long long)above) : (64 - lane);
        atomicAdd(frame_count + lin, run);
    }
}

__global__ void __launch_bounds__(DIF_BLOCK) k_voxel_count(Geo g, const float* __restrict__ xyz, int64_t N, int* __restrict__ pt_lin,
                                                         int* __restrict__ frame_count, int* __restrict__ counters, int px_lo, int px_hi) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // grid covers N rounded up to a wave
    const bool in = i < N;
    voxel_count_point(g, in, in ? xyz[i * 3 + 0] : 0.f, in ? xyz[i * 3 + 1] : 0.f, in ? xyz[i * 3 + 2] : 0.f, i, pt_lin, frame_count, counters, px_lo, px_hi);
}

// a1 + a2 + a3 in one pass for a streaming caller: depth pixel -> world point and normal (written out for the later stages) -> voxel id
// and per-voxel count, without re-reading the points.
struct ImageGeo { int H, W; float fx, fy, cx, cy; };

// Where the stages behind the first kernel get a point's world coordinates and normal from: the (N,3) arrays (written by the first
// kernel or supplied by the caller), or — arrays NULL, a streaming frame — recomputed from the depth pixel and the pose with the very
// operations of the first kernel (unproject_point: same order, no contraction => the same bits).  A frame's ~3 % of points that pass
// the focus test and its gathered encoder rows are all that ever need them: writing 24 bytes per pixel for every pixel of every frame
// (7.4 MB at 640x480) just to read a few per cent back was the first kernel's largest cost.
struct PointSrc {
    const float* xyz; const float* normal;
    const dif_frame_t* frame; ImageGeo im;
};

__device__ __forceinline__ Pose pose_of(const dif_frame_t* __restrict__ frame) {
    Pose P;
#pragma unroll
    for (int k = 0; k < 9; ++k) P.r[k] = frame->pose[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) P.t[k] = frame->pose[9 + k];
    return P;
}

__device__ __forceinline__ void src_point(const PointSrc& s, const Pose& P, int64_t i, float (&p)[3]) {
    if (s.xyz) { p[0] = s.xyz[i * 3 + 0]; p[1] = s.xyz[i * 3 + 1]; p[2] = s.xyz[i * 3 + 2]; return; }
    float nv[3];
    unproject_point(s.frame->depth, nullptr, i, s.im.W, s.im.fx, s.im.fy, s.im.cx, s.im.cy, P, p, nv);
}

__device__ __forceinline__ void src_point_normal(const PointSrc& s, const Pose& P, int64_t i, float (&p)[3], float (&nv)[3]) {
    if (s.xyz) {
        p[0] = s.xyz[i * 3 + 0]; p[1] = s.xyz[i * 3 + 1]; p[2] = s.xyz[i * 3 + 2];
        nv[0] = s.normal[i * 3 + 0]; nv[1] = s.normal[i * 3 + 1]; nv[2] = s.normal[i * 3 + 2];
        return;
    }
    unproject_point(s.frame->depth, s.frame->normal_cam, i, s.im.W, s.im.fx, s.im.fy, s.im.cx, s.im.cy, P, p, nv);
}

__device__ __forceinline__ void unproject_voxel_count_body(const Geo& g, const dif_frame_t* __restrict__ frame, int H, int W, float fx, float fy,
                                                           float cx, float cy, float* __restrict__ xyz, float* __restrict__ nrm,
                                                           int* __restrict__ pt_lin, int* __restrict__ frame_count, int* __restrict__ counters,
                                                           int px_lo, int px_hi, const dif_pending_export_t* __restrict__ pending, int nb_x,
                                                           dif_frame_t* __restrict__ frame_copy, uint8_t* __restrict__ chunk_any) {
    // The first nb_x workgroups (dispatched first, so that the copy runs beside the whole point pass and not at its tail) carry out a third of
    // the previous extract's deferred triangle export; the other two thirds ride with the next two kernels.
    if ((int)blockIdx.x < nb_x) {
        export_pending_rows(pending, (int)blockIdx.x, 3 * nb_x);
        return;
    }
    const int64_t N = (int64_t)H * W;
    const int64_t i = (int64_t)((int)blockIdx.x - nb_x) * blockDim.x + threadIdx.x;
    const bool in = i < N;
    if ((int)blockIdx.x == nb_x && threadIdx.x < 16 && frame_copy)       // the descriptor, for the later kernels of this frame
        reinterpret_cast<uint32_t*>(frame_copy)[threadIdx.x] = reinterpret_cast<const uint32_t*>(frame)[threadIdx.x];
    float p[3] = {0.f, 0.f, 0.f}, nv[3];
    if (in) {
        const Pose P = pose_of(frame);
        if (xyz) {
            unproject_point(frame->depth, frame->normal_cam, i, W, fx, fy, cx, cy, P, p, nv);
            xyz[i * 3 + 0] = p[0]; xyz[i * 3 + 1] = p[1]; xyz[i * 3 + 2] = p[2];
            nrm[i * 3 + 0] = nv[0]; nrm[i * 3 + 1] = nv[1]; nrm[i * 3 + 2] = nv[2];
        } else {
            unproject_point(frame->depth, nullptr, i, W, fx, fy, cx, cy, P, p, nv);      // (the normals are not needed before the encoder)
        }
    }
    voxel_count_point(g, in, p[0], p[1], p[2], i, pt_lin, frame_count, counters, px_lo, px_hi);
    if (chunk_any) {        // spatial tiling: does this piece of the frame hold any point inside the slab + halo?  (what it wrote to pt_lin says so)
        const int any = __syncthreads_or((int)(in && pt_lin[i] >= 0));
        if (threadIdx.x == 0) chunk_any[(int)blockIdx.x - nb_x] = (uint8_t)(any != 0);
    }
}

struct UvcArgs {            // per map; the image geometry is shared by the maps of a batched launch
    Geo g; const dif_frame_t* frame; float* xyz; float* nrm; int* pt_lin; int* frame_count; int* counters; int px_lo, px_hi;
    const dif_pending_export_t* pending;
    dif_frame_t* frame_copy;        // device copy of the descriptor for the later kernels of the frame (the caller's may sit in pinned host memory)
    uint8_t* chunk_any;             // spatial tiling: per workgroup of this kernel, "holds a point inside the slab + halo" (NULL: not kept)
};

__global__ void __launch_bounds__(DIF_BLOCK) k_unproject_voxel_count(UvcArgs a, ImageGeo im, int nb_x) {
    unproject_voxel_count_body(a.g, a.frame, im.H, im.W, im.fx, im.fy, im.cx, im.cy, a.xyz, a.nrm, a.pt_lin, a.frame_count, a.counters, a.px_lo, a.px_hi,
                               a.pending, nb_x, a.frame_copy, a.chunk_any);
}
__global__ void __launch_bounds__(DIF_BLOCK) k_unproject_voxel_count_batch(Batch<UvcArgs> b, ImageGeo im, int nb_x) {
    const UvcArgs& a = b.s[blockIdx.y];
    unproject_voxel_count_body(a.g, a.frame, im.H, im.W, im.fx, im.fy, im.cx, im.cy, a.xyz, a.nrm, a.pt_lin, a.frame_count, a.counters, a.px_lo, a.px_hi,
                               a.pending, nb_x, a.frame_copy, a.chunk_any);
}

// K2: prune mask + candidate voxels.  mask[i] = count(voxel of i) > prune_min_vox_obs (map.py:375).  A kept point whose
// voxel has no slot marks that voxel and its 6 clamped neighbours (if empty) in the bitmap (map.py:383-386).
__device__ __forceinline__ void prune_mark_body(const Geo& g, int prune_min, const int* __restrict__ pt_lin, int64_t N,
                                                const int* __restrict__ frame_count, const int64_t* __restrict__ indexer,
                                                uint8_t* __restrict__ unq_mask, const GridMarks& marks,
                                                int* __restrict__ counters, const dif_pending_export_t* __restrict__ pending, int nb_x,
                                                const uint8_t* __restrict__ chunk_any) {
    if ((int)blockIdx.x < nb_x) {           // leading workgroups: the second third of a deferred triangle export
        export_pending_rows(pending, nb_x + (int)blockIdx.x, 3 * nb_x);
        return;
    }
    const uint32_t* bits = marks.bits;
    int64_t i = (int64_t)((int)blockIdx.x - nb_x) * blockDim.x + threadIdx.x;
    if (chunk_any && !chunk_any[(int)blockIdx.x - nb_x]) {          // spatial tiling: no point of this piece lies near the slab — nothing is kept
        if (i < N) unq_mask[i] = 0;
        return;
    }
    int lane = lane_id();
    int lin = (i < N) ? pt_lin[i] : -2;
    bool keep = false;
    const int cnt = (lin >= 0) ? frame_count[lin] : 0;            // both look-ups hang off `lin` only: issued together
    const int64_t own = (lin >= 0) ? indexer[lin] : 0;
    if (lin >= 0) keep = (prune_min > 0) ? (cnt > prune_min) : true;
    if (i < N) unq_mask[i] = keep ? 1 : 0;
    int prev = __shfl_up(lin, 1);
    bool head = (lane == 0) || (prev != lin);
    if (head && keep && own == -1) {
        int ix, iy, iz;
        unlinearize(g, lin, ix, iy, iz);
        int cand[7];
        cand[0] = lin;
        cand[1] = linearize(g, clampi(ix - 1, 0, g.nx - 1), iy, iz);
        cand[2] = linearize(g, clampi(ix + 1, 0, g.nx - 1), iy, iz);
        cand[3] = linearize(g, ix, clampi(iy - 1, 0, g.ny - 1), iz);
        cand[4] = linearize(g, ix, clampi(iy + 1, 0, g.ny - 1), iz);
        cand[5] = linearize(g, ix, iy, clampi(iz - 1, 0, g.nz - 1));
        cand[6] = linearize(g, ix, iy, clampi(iz + 1, 0, g.nz - 1));
        // look-ups, bitmap updates and the first setters' scan-block counts each go out as one batch
        bool empty[7];
#pragma unroll
        for (int c = 0; c < 7; ++c) empty[c] = indexer[cand[c]] == -1;
        uint32_t prev[7];
#pragma unroll
        for (int c = 0; c < 7; ++c) {
            const uint32_t b = 1u << (cand[c] & 31);
            prev[c] = (!empty[c] || (bits[cand[c] >> 5] & b)) ? 0xFFFFFFFFu : atomicOr(marks.bits + (cand[c] >> 5), b);
        }
#pragma unroll
        for (int c = 0; c < 7; ++c)
            if (!(prev[c] & (1u << (cand[c] & 31)))) atomicAdd(marks.tot + (cand[c] >> 5) / marks.per_words, 1);
    }
}

struct PruneArgs {
    Geo g; int prune_min; const int* pt_lin; const int* frame_count; const int64_t* indexer; uint8_t* unq_mask; GridMarks marks; int* counters;
    const dif_pending_export_t* pending; const uint8_t* chunk_any;
};

__global__ void __launch_bounds__(DIF_BLOCK) k_prune_mark(PruneArgs a, int64_t N, int nb_x) {
    prune_mark_body(a.g, a.prune_min, a.pt_lin, N, a.frame_count, a.indexer, a.unq_mask, a.marks, a.counters, a.pending, nb_x, a.chunk_any);
}
__global__ void __launch_bounds__(DIF_BLOCK) k_prune_mark_batch(Batch<PruneArgs> b, int64_t N, int nb_x) {
    const PruneArgs& a = b.s[blockIdx.y];
    prune_mark_body(a.g, a.prune_min, a.pt_lin, N, a.frame_count, a.indexer, a.unq_mask, a.marks, a.counters, a.pending, nb_x, a.chunk_any);
}

// K3: ordered compaction of the candidate bitmap -> slots n_occupied, n_occupied+1, ... in ASCENDING lin order
// (torch.unique order, map.py:385-387, 310-319).  Clears the bitmap as it goes.
struct AllocFunctor {
    uint32_t* bits;
    int64_t* indexer;
    int64_t* pos;
    int* counters;
    int64_t capacity;
    HaloLists hl;           // spatial tiling: a new owned boundary voxel goes into the frame's halo delta
    __device__ int count(int w) const { return __popc(bits[w]); }
    __device__ void emit(int w, int offset) const {
        uint32_t word = bits[w];
        bits[w] = 0u;
        int base = counters[DIF_C_N_OCCUPIED] + offset;
        while (word) {
            int b = __ffs((int)word) - 1;
            word &= word - 1;
            int lin = w * 32 + b;
            if (base < capacity) {
                indexer[lin] = base;
                pos[base] = lin;
                hl.note(base, lin);
            }
            ++base;
        }
    }
    __device__ void finish(int total) const { counters[DIF_C_ALLOC_NEW] = total; }
};

// K4: (i) commit n_occupied += newly allocated (all pass-2 blocks of K3 have read the old value by now),
// (ii) restore frame_count to zero, (iii) focus mask + 8-offset gather (map.py:389-433), written as a COMPACTED list of the valid
// (offset o, point i) pairs: entry = (slot of the neighbour voxel, o*N + i) for every pair whose neighbour voxel is in the encode set
// {obs_count < encoder_count_th}.  In steady state ~3 % of the 8N pairs are valid, so nothing 8N-sized is written or read again.
// A workgroup's pairs are GROUPED BY SLOT before they are written (a counting sort through a 256-entry LDS hash table: the pairs
// of a 16 x 16 pixel tile hit 10-30 voxels), so every (workgroup, slot) combination is one contiguous run of the list — which is
// what keeps the number of run records per voxel small in k_encode / k_fuse.  Which list range a workgroup gets is decided by one
// atomic per workgroup, and the order inside a run by LDS atomics; neither affects results (per-voxel sums are exact integer sums).
__device__ __forceinline__ bool in_encode_set(int64_t slot, const float* __restrict__ obs, float th) { return slot >= 0 && obs[slot] < th; }

#define FG_TABLE 256            /* == DIF_BLOCK: one table entry per thread in the scan below */
__device__ __forceinline__ void focus_gather_body(const Geo& g, float enc_th, const PointSrc& src, const int* __restrict__ pt_lin,
                                                  const uint8_t* __restrict__ unq_mask, int64_t N, int* __restrict__ frame_count,
                                                  const int64_t* __restrict__ indexer, const float* __restrict__ obs,
                                                  uint2* __restrict__ pair_list, int* __restrict__ counters, int64_t capacity, int img_w,
                                                  int* __restrict__ grid_tot, int own_lo, int own_hi,
                                                  const dif_pending_export_t* __restrict__ pending, int nb_x, const uint8_t* __restrict__ chunk_any) {
    if ((int)blockIdx.x < nb_x) {           // leading workgroups: the last third of a deferred triangle export
        export_pending_rows(pending, 2 * nb_x + (int)blockIdx.x, 3 * nb_x);
        return;
    }
    // (measured: neighbouring tiles dealt to ONE XCD, common.hip.h's xcd_run_item, make this kernel 0.3 us slower, not faster — 12.55 against 12.27 us)
    const int bid = (int)blockIdx.x - nb_x;
    __shared__ unsigned tkey[FG_TABLE];
    if (bid == 0)                                            // the allocation scan has consumed the bitmap's block totals: back to idle 0
        for (int t = (int)threadIdx.x; t < 1024; t += DIF_BLOCK) grid_tot[t] = 0;
    __shared__ int tcnt[FG_TABLE];        // rows per table entry, then (after the scan) the entry's first list position within the workgroup
    __shared__ int smem[8];
    __shared__ int s_loose, s_base;
    tkey[threadIdx.x] = DIF_INVALID_KEY;
    tcnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_loose = 0;
    // Points of a frame (img_w > 0, N = H * img_w with both multiples of 16) are walked in 16 x 16 pixel tiles: a workgroup then
    // touches 10-30 voxels instead of the ~90 that a 256-pixel piece of an image row does.
    int64_t i = (int64_t)bid * blockDim.x + threadIdx.x;      // grid covers N rounded up to whole workgroups
    if (img_w > 0) {
        const int tiles_x = img_w >> 4;
        const int ty = bid / tiles_x, tx = bid % tiles_x;
        i = (int64_t)(ty * 16 + (int)(threadIdx.x >> 4)) * img_w + tx * 16 + (int)(threadIdx.x & 15);
    }
    if (i == 0) {
        int n = counters[DIF_C_N_OCCUPIED] + counters[DIF_C_ALLOC_NEW];
        if (n > capacity) { n = (int)capacity; counters[DIF_C_OVERFLOW] = 1; }
        counters[DIF_C_N_OCCUPIED] = n;
    }
    // spatial tiling: a 16 x 16 pixel tile none of whose 16 row pieces holds a point near the slab has nothing to gather (and no frame count to
    // restore: nothing of it was counted) — one byte per row piece instead of 5 bytes per pixel
    if (chunk_any && img_w > 0) {
        int flag = 0;
        if (threadIdx.x < 16) flag = chunk_any[(i + (int64_t)threadIdx.x * img_w) >> 8];      // thread t: the piece that holds row t of the tile (pieces are 256 points)
        if (!__syncthreads_or(flag)) return;
    }
    const int lin = (i < N) ? pt_lin[i] : -1;
    uint32_t key[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) key[o] = DIF_INVALID_KEY;
    // get_pruned_surface: own voxel in expand(encode set) <=> own voxel or an in-grid 6-neighbour is in the set.  That is a property of the
    // VOXEL, and neighbouring pixels share voxels: the first lane of every run of equal ids does the seven look-ups, the rest of the run
    // takes its answer (in steady state ~97 % of the points fail this test, so it is most of the kernel's gathers).
    const int lane = lane_id(), wid = (int)(threadIdx.x >> 6);
    const bool kept = lin >= 0 && unq_mask[i];
    if (lin >= 0) frame_count[lin] = 0;
    const int prev = __shfl_up(lin, 1);
    const bool head = (lane == 0) || (prev != lin);
    const unsigned long long heads = __ballot(head);
    bool focus = false;
    if (head && kept) {
        int ix, iy, iz;
        unlinearize(g, lin, ix, iy, iz);
        // all seven indexer entries first, then all seven observation counts: two dependent loads deep instead of fourteen
        const int plane = g.ny * g.nz;
        const int nl[7] = {lin, lin - plane, lin + plane, lin - g.nz, lin + g.nz, lin - 1, lin + 1};
        const bool in_grid[7] = {true, ix > 0, ix < g.nx - 1, iy > 0, iy < g.ny - 1, iz > 0, iz < g.nz - 1};
        int64_t slot[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) slot[q] = in_grid[q] ? indexer[nl[q]] : -1;
        float w[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) w[q] = slot[q] >= 0 ? obs[slot[q]] : enc_th;
#pragma unroll
        for (int q = 0; q < 7; ++q) focus |= w[q] < enc_th;
    }
    const int my_head = 63 - __clzll((long long)(heads & ((2ull << lane) - 1ull)));      // nearest run start at or below this lane
    focus = __shfl((int)focus, my_head) != 0;
    if (kept && focus) {
        float xn, yn, zn; int ix, iy, iz;
        float p[3];
        src_point(src, src.xyz ? Pose{} : pose_of(src.frame), i, p);
        voxel_of(g, p[0], p[1], p[2], xn, yn, zn, ix, iy, iz);
        int64_t slot[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            float ox = (o & 4) ? 0.5f : -0.5f, oy = (o & 2) ? 0.5f : -0.5f, oz = (o & 1) ? 0.5f : -0.5f;   // map.py:186-189
            int gx = clampi((int)(ceilf(xn + ox) - 1.0f), 0, g.nx - 1);                                   // map.py:422-424
            int gy = clampi((int)(ceilf(yn + oy) - 1.0f), 0, g.ny - 1);
            int gz = clampi((int)(ceilf(zn + oz) - 1.0f), 0, g.nz - 1);
            // spatial tiling: a voxel is only ever written by its owner (the others receive it with the halo refresh), so pairs that
            // target a voxel outside the own slab are dropped here — no encoder rows are spent on them
            slot[o] = (gx >= own_lo && gx < own_hi) ? indexer[linearize(g, gx, gy, gz)] : -1;
        }
        float w[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) w[o] = slot[o] >= 0 ? obs[slot[o]] : enc_th;      // all eight counts in flight together
#pragma unroll
        for (int o = 0; o < 8; ++o)
            if (w[o] < enc_th) key[o] = (uint32_t)slot[o];
    }
    // ---- group by slot inside the workgroup, then one list reservation per workgroup ----
    bool any = false;
#pragma unroll
    for (int o = 0; o < 8; ++o) any |= key[o] != DIF_INVALID_KEY;
    if (!__syncthreads_or((int)any)) return;                   // (also orders the table initialisation before its first use)
    int where[8], rank[8];                                       // table entry (-1: table full, "loose" row) and rank inside it
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        where[o] = -1; rank[o] = 0;
        if (key[o] == DIF_INVALID_KEY) continue;
        unsigned h = (key[o] * 2654435761u) >> 24;
        for (int probes = 0; probes < FG_TABLE; ++probes) {
            const unsigned old = atomicCAS(tkey + h, DIF_INVALID_KEY, key[o]);
            if (old == DIF_INVALID_KEY || old == key[o]) { where[o] = (int)h; break; }
            h = (h + 1) & (FG_TABLE - 1);
        }
        rank[o] = atomicAdd(where[o] >= 0 ? tcnt + where[o] : &s_loose, 1);      // > 256 voxels in one workgroup (scattered points): ungrouped tail
    }
    __syncthreads();
    int grouped;
    const int first = block_excl_scan(tcnt[threadIdx.x], smem, grouped);
    __syncthreads();
    tcnt[threadIdx.x] = first;
    if (threadIdx.x == 0) s_base = atomicAdd(counters + DIF_C_M, grouped + s_loose);      // M of map.py:434-435
    __syncthreads();
#pragma unroll
    for (int o = 0; o < 8; ++o)
        if (key[o] != DIF_INVALID_KEY)
            pair_list[s_base + (where[o] >= 0 ? tcnt[where[o]] : grouped) + rank[o]] = make_uint2(key[o], (uint32_t)((int64_t)o * N + i));
}

struct GatherArgs {
    Geo g; float enc_th; PointSrc src; const int* pt_lin; const uint8_t* unq_mask; int* frame_count; const int64_t* indexer; const float* obs;
    uint2* pair_list; int* counters; int64_t capacity; int* grid_tot; int own_lo, own_hi; const dif_pending_export_t* pending; const uint8_t* chunk_any;
};

__global__ void __launch_bounds__(DIF_BLOCK) k_focus_gather(GatherArgs a, int64_t N, int img_w, int nb_x) {
    focus_gather_body(a.g, a.enc_th, a.src, a.pt_lin, a.unq_mask, N, a.frame_count, a.indexer, a.obs, a.pair_list, a.counters, a.capacity, img_w, a.grid_tot,
                      a.own_lo, a.own_hi, a.pending, nb_x, a.chunk_any);
}
__global__ void __launch_bounds__(DIF_BLOCK) k_focus_gather_batch(Batch<GatherArgs> b, int64_t N, int img_w, int nb_x) {
    const GatherArgs& a = b.s[blockIdx.y];
    focus_gather_body(a.g, a.enc_th, a.src, a.pt_lin, a.unq_mask, N, a.frame_count, a.indexer, a.obs, a.pair_list, a.counters, a.capacity, img_w, a.grid_tot,
                      a.own_lo, a.own_hi, a.pending, nb_x, a.chunk_any);
}

// =================================================================================================================
// a7..a9 : gather + encoder (MFMA) + per-voxel sums
// =================================================================================================================
// Persistent: one 512-thread workgroup per CU keeps the 107 KB of packed encoder weights in LDS; each wave pulls tiles of 32
// CONSECUTIVE list entries (every tile full: no per-voxel padding, no sort), runs them through the MFMA chain and reduces the 29
// output features over each RUN of equal slots inside the tile with one segmented wave scan.  Every run leaves a 256-byte record
// (32 x int64: the 29 feature sums, the run length in the spare position 29) at a fixed place, rec[tile*32 + rank of the run], and
// enters its slot's record DIRECTORY with one atomic (rec_dir[slot][16]: word 0 = number of records, idle 0; words 2..15 = the
// first 14 record ids, which k_fuse fetches in parallel; further ones are chained through word 1 / rec_next).  The first run of a
// slot also appends the slot to the frame's update list.
// The sums are kept in 2^-30 FIXED POINT (int64): integer addition is associative, so the per-voxel sum does not depend on list
// order, tile boundaries or chain order => bit-reproducible, and more accurate than an fp32 running sum (the reference sums with
// float atomics in arbitrary order, indexing.cu:59-71).
#define DIF_FIX_SCALE 1073741824.0f          /* 2^30: |enc| < 2^12 and < 2^21 rows per voxel keep the sum inside int64 */
#define DIF_REC_WORDS 32                     /* int64 per record */
#define DIF_REC_COUNT_POS 29                 /* record position that carries the run length (feature 29 is a zero row of the MFMA tile) */
#define DIF_DIR_WORDS 16                     /* int32 per slot directory: count | overflow chain head | 14 record ids */
#define DIF_DIR_IDS (DIF_DIR_WORDS - 2)

#ifdef DIF_TRACE            // tools/trace_decode.py --encode: per-wave phase timestamps (100 MHz wall clock) of the last k_encode launch
__device__ unsigned long long g_en_trace[4096 * 8];       // (up to 16 waves x 256 workgroups; the x6 kernel runs 12 waves per workgroup)
#define EN_STAMP(slot) do { if (lane_id() == 0) g_en_trace[((threadIdx.x >> 6) * gridDim.x + blockIdx.x) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define EN_STAMP(slot) do { } while (0)
#endif

// X6: the tile runs on the bf16 matrix pipe (encoder_tile_x6; wblob = packing.py:pack_encoder_x6).
// NS > 1: the tiles of S <= NS maps are walked as ONE range (map 0's tiles, then map 1's, ...): the weights are staged once per
// workgroup and the launch carries S frames' worth of tiles per SIMD.  Which map a tile belongs to is wave-uniform (scalar loads of
// that map's pointers from the kernel-argument array); per map nothing changes — same tiles, same records, same directory.
struct EncArgs {
    Geo g; PointSrc src; const uint2* pair_list; int* rec_dir; int* rec_next; long long* rec; int* upd_list; int* counters;
    uint8_t* dirty; int* dirty_tot;
};
#ifndef ENC_X6_THREADS
#define ENC_X6_THREADS 768        /* twelve waves per CU: 3,072 tile slots on the chip */
#endif
template <bool X6, int NS>
__device__ __forceinline__ void encode_body(const BatchN<EncArgs, NS>& B, int S, const float* __restrict__ wblob, int64_t N) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    EN_STAMP(0);
    const int lane = lane_id(), half = lane >> 5, col = lane & 31;
    // tile t goes to wave (t / #blocks) of block (t % #blocks): a partly filled launch spreads over all CUs and SIMDs first
    const int wave = __builtin_amdgcn_readfirstlane((int)((threadIdx.x >> 6) * gridDim.x + blockIdx.x));
    const int nwaves = (int)(gridDim.x * (blockDim.x >> 6));
    Ranges<NS, int> rg;                   // cnt: gathered rows of map j; pre: its first tile in the concatenated range (maps beyond S: empty)
    rg.pre[0] = 0;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        rg.cnt[j] = (j < S) ? B.s[j].counters[DIF_C_M] : 0;
        rg.pre[j + 1] = rg.pre[j] + ((rg.cnt[j] + 31) >> 5);
    }
    const int n_tiles = rg.total();
    // a tile's inputs: its 32 list entries and, through them, the points' coordinates and normals (two dependent gathers).  They are
    // requested one tile ahead: the first tile's before the weights are staged into LDS, the next tile's before the current tile's MFMA
    // chain — neither wait is on the critical path any more
    struct TileIn { uint2 e; float x0, x1, x2; };
    // (sm, lt, M: the tile's map, its index in that map and the map's row count — located by the caller, so that the range tables never
    // travel through a closure and stay in scalar registers)
    auto gather = [&](bool any, int sm, int lt, int M) __attribute__((always_inline)) {
        TileIn t;
        t.e = make_uint2(DIF_INVALID_KEY, 0u);
        t.x0 = t.x1 = t.x2 = 0.f;
        if (any) {
            const EncArgs& a = B.s[sm];
            const Geo g = a.g;
            const int row = lt * 32 + col;
            if (row < M) {
                t.e = a.pair_list[row];
                const uint32_t v = t.e.y;
                int o = 0;
#pragma unroll
                for (int k = 1; k < 8; ++k) o += ((int64_t)v >= (int64_t)k * N) ? 1 : 0;
                int64_t i = (int64_t)v - (int64_t)o * N;
                float pw[3], nw[3];
                src_point_normal(a.src, a.src.xyz ? Pose{} : pose_of(a.src.frame), i, pw, nw);
                float xn = normalize1(pw[0], g.bx, g.vs);
                float yn = normalize1(pw[1], g.by, g.vs);
                float zn = normalize1(pw[2], g.bz, g.vs);
                float ox = (o & 4) ? 0.5f : -0.5f, oy = (o & 2) ? 0.5f : -0.5f, oz = (o & 1) ? 0.5f : -0.5f;
                // (the clamp in integers — `ceilf(..) - 1` is integer-valued and, for the points of the list, inside the grid's range — with the
                // bounds in scalar registers: as floats, (float)(n - 1) are loop-invariant VALU results that hipcc kept in three vector registers
                // across the tile loop of the bf16-pipe kernel, over its 168-register budget: three scratch reloads per tile until round 5)
                float gx = (float)clampi((int)(ceilf(xn + ox) - 1.0f), 0, g.nx - 1);
                float gy = (float)clampi((int)(ceilf(yn + oy) - 1.0f), 0, g.ny - 1);
                float gz = (float)clampi((int)(ceilf(zn + oz) - 1.0f), 0, g.nz - 1);
                float rx = (xn - gx) - 0.5f, ry = (yn - gy) - 0.5f, rz = (zn - gz) - 0.5f;      // map.py:425
                float nxv = nw[0], nyv = nw[1], nzv = nw[2];
                t.x0 = half ? ry : rx;
                t.x1 = half ? nxv : rz;
                t.x2 = half ? nzv : nyv;
            }
        }
        return t;
    };
    int sm_n, lt_n, M_n;
    rg.locate(wave, sm_n, lt_n, M_n);
    TileIn nxt = gather(wave < n_tiles, sm_n, lt_n, M_n);
    __builtin_amdgcn_sched_barrier(0);
    stage_weights(lds, wblob, X6 ? E6_BYTES / 4 : ENC_FLOATS);
    EN_STAMP(1);
    for (int T = wave; T < n_tiles; T += nwaves) {
        const TileIn cur = nxt;
        const int sm = sm_n, tile = lt_n, M = M_n;
        rg.locate(T + nwaves, sm_n, lt_n, M_n);
        nxt = gather(T + nwaves < n_tiles, sm_n, lt_n, M_n);
        const EncArgs& a = B.s[sm];
        const int row = tile * 32 + col;
        const bool live = row < M;
        const uint2 e = cur.e;
        const uint32_t key = e.x;
        // runs of equal slots (both halves of the wave hold the same 32 rows and compute the same run structure)
        const uint32_t key_prev = (uint32_t)__shfl_up((int)key, 1), key_next = (uint32_t)__shfl_down((int)key, 1);
        const bool run_head = (col == 0) || (key_prev != key);
        const bool run_tail = (col == 31) || (key_next != key);
        const uint32_t heads32 = (uint32_t)(__ballot(run_head) >> (half * 32));
        const int my_head = 31 - __clz((int)(heads32 & ((2u << col) - 1u)));      // column where this lane's run starts
        const int rec_id = tile * 32 + __popc(heads32 & ((2u << col) - 1u)) - 1;  // record of this lane's run
        const float x0 = cur.x0, x1 = cur.x1, x2 = cur.x2;
        // enter the run in its slot's directory.  Asked after the gathers above have been consumed and before the MFMA chain, answered
        // after it: the round trip hides behind ~11 us of matrix work.
        EN_STAMP(2);
        int dir_pos = 0;
        const bool pusher = live && run_tail && half == 0;
        int* dir = a.rec_dir + (int64_t)key * DIF_DIR_WORDS;
        if (pusher) dir_pos = atomicAdd(dir, 1);
        f16v out;
        if constexpr (X6) out = encoder_tile_x6(lds, x0, x1, x2, lane);
        else out = encoder_tile(lds, x0, x1, x2, lane);
        EN_STAMP(3);
        long long* p = a.rec + (int64_t)rec_id * DIF_REC_WORDS + half * 16;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            long long v = live ? __float2ll_rn(out[r] * DIF_FIX_SCALE) : 0ll;
            if (half == 1 && r == 13) v = live ? 1ll : 0ll;          // feature 29 (a zero row): carries the run length instead
            v = seg_incl_scan32(v, col, my_head);            // sum of the lane's run up to its column
            if (live && run_tail) p[r] = v;
        }
        if (pusher) {
            if (dir_pos < DIF_DIR_IDS) dir[2 + dir_pos] = rec_id;
            else a.rec_next[rec_id] = atomicExch(dir + 1, rec_id + 1);                      // a voxel fed by many workgroups: chained
            if (dir_pos == 0) {                                                             // first run of this slot in the frame (C of map.py:437)
                a.upd_list[atomicAdd(a.counters + DIF_C_C, 1)] = (int)key;
                // k_fuse will set the slot's dirty flag: if it is not set yet, count it into the total of its 256-slot block here (extract's
                // ordered compaction of the dirty set then needs no counting pass) — one lane per updated slot and frame, off k_fuse's tail
                if (a.dirty_tot && !a.dirty[key]) atomicAdd(a.dirty_tot + (key >> 8), 1);
            }
        }
        EN_STAMP(4);
    }
    EN_STAMP(5);
}

// One map: the pointers arrive as noalias kernel arguments (the body's accesses keep that provenance).
template <bool X6>
__global__ void __launch_bounds__(X6 ? ENC_X6_THREADS : 512, X6 ? 1 : 2)
k_encode(Geo g, const float* __restrict__ wblob, const float* __restrict__ xyz, const float* __restrict__ normal, const dif_frame_t* __restrict__ frame,
         ImageGeo im, int64_t N, const uint2* __restrict__ pair_list, int* __restrict__ rec_dir, int* __restrict__ rec_next, long long* __restrict__ rec,
         int* __restrict__ upd_list, int* __restrict__ counters, uint8_t* __restrict__ dirty, int* __restrict__ dirty_tot) {
    const BatchN<EncArgs, 1> B{{EncArgs{g, PointSrc{xyz, normal, frame, im}, pair_list, rec_dir, rec_next, rec, upd_list, counters, dirty, dirty_tot}}};
    encode_body<X6, 1>(B, 1, wblob, N);
}

template <bool X6>
__global__ void __launch_bounds__(X6 ? ENC_X6_THREADS : 512, X6 ? 1 : 2)
k_encode_batch(Batch<EncArgs> B, int S, const float* __restrict__ wblob, int64_t N) {
    encode_body<X6, DIF_MAX_STREAMS>(B, S, wblob, N);
}

// encoder on explicit rows (flat op / tests): per-row outputs instead of per-voxel sums, so a dedicated small kernel
template <bool X6>
__global__ void __launch_bounds__(512, X6 ? 1 : 2) k_encode_rows(const float* __restrict__ wblob, const float* __restrict__ rows, int64_t n, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    stage_weights(lds, wblob, X6 ? E6_BYTES / 4 : ENC_FLOATS);
    const int lane = lane_id(), half = lane >> 5, col = lane & 31;
    // work item w goes to wave (w / #blocks) of block (w % #blocks): a partly filled launch spreads over all CUs and SIMDs first
    const int wave = (int)((threadIdx.x >> 6) * gridDim.x + blockIdx.x);
    const int nwaves = (int)(gridDim.x * (blockDim.x >> 6));
    const int64_t n_tiles = (n + 31) / 32;
    for (int64_t tile = wave; tile < n_tiles; tile += nwaves) {
        int64_t row = tile * 32 + col;
        bool live = row < n;
        float x0 = 0.f, x1 = 0.f, x2 = 0.f;
        if (live) {
            const float* p = rows + row * 6;
            x0 = half ? p[1] : p[0];
            x1 = half ? p[3] : p[2];
            x2 = half ? p[5] : p[4];
        }
        f16v o;
        if constexpr (X6) o = encoder_tile_x6(lds, x0, x1, x2, lane);
        else o = encoder_tile(lds, x0, x1, x2, lane);
        if (live) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int f = (r & 3) + 8 * (r >> 2) + 4 * half;
                if (f < L) out[row * L + f] = o[r];
            }
        }
    }
}

// a10: fusion update (map.py:448-452).  One 32-lane group per updated slot: sum the slot's run records (directory entries fetched
// together, overflow chain walked), fuse, return the directory to its idle state.
__device__ __forceinline__ void fuse_body(const long long* __restrict__ rec, const int* __restrict__ rec_next, int* __restrict__ rec_dir,
                                          const int* __restrict__ upd_list, float* __restrict__ latent, float* __restrict__ obs,
                                          uint8_t* __restrict__ dirty, int* __restrict__ counters, const int64_t* __restrict__ slot_lin, const HaloLists& hl,
                                          dif_pending_export_t* __restrict__ pending, int* __restrict__ dirty_tot, int* __restrict__ fc) {
    if (pending && blockIdx.x == 0 && threadIdx.x == 0) {       // the point kernels' extra workgroups have done the copy (kernels ago: complete)
        pending->pending = 0;
        if (pending->notify) {              // tell the host without an event in the queue (dif_extract_buffers_t.export_notify)
            __threadfence_system();         // (the copy's stores left with earlier kernels; nothing of this frame may overtake them on the way out)
            *pending->notify = pending->seq;
        }
    }
    const int n_upd = counters[DIF_C_C];
    const int first_new = counters[DIF_C_N_OCCUPIED] - counters[DIF_C_ALLOC_NEW];      // this frame's new slots are already in the halo delta
    const int grp = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 5), ngrp = (int)((gridDim.x * blockDim.x) >> 5);
    const int f = threadIdx.x & 31;
    // where feature f sits in a record: the accumulator-fragment order of the tile's two halves (mlp.hip.h)
    const int pos = ((f >> 2) & 1) * 16 + (f & 3) + 4 * (f >> 3);
    for (int u = grp; u < n_upd; u += ngrp) {
        const int s = upd_list[u];
        int* dir = rec_dir + (int64_t)s * DIF_DIR_WORDS;
        const int n_rec = dir[0];
        const int my_id = (f < DIF_DIR_IDS && f < n_rec) ? dir[2 + f] : -1;       // lane f fetches directory entry f
        const float w_old = obs[s];                                               // (requested beside the directory, not behind the records)
        const float z_old = f < L ? latent[(int64_t)s * L + f] : 0.0f;
        long long Si = 0;
        int cnt = 0;
        const int n_dir = n_rec < DIF_DIR_IDS ? n_rec : DIF_DIR_IDS;
        for (int j = 0; j < n_dir; ++j) {                       // independent loads: all in flight together
            const long long* r = rec + (int64_t)__shfl(my_id, j, 32) * DIF_REC_WORDS;
            Si += r[pos];
            cnt += (int)r[DIF_REC_COUNT_POS];
        }
        if (n_rec > DIF_DIR_IDS)
            for (int id = dir[1]; id != 0; id = rec_next[id - 1]) {
                const long long* r = rec + (int64_t)(id - 1) * DIF_REC_WORDS;
                Si += r[pos];
                cnt += (int)r[DIF_REC_COUNT_POS];
            }
        if (f < L) {
            float S = (float)Si * (1.0f / DIF_FIX_SCALE);    // one rounding: exact integer sum -> nearest float
            S = S + z_old * w_old;                           // map.py:449
            float w_new = w_old + (float)cnt;                // map.py:450
            latent[(int64_t)s * L + f] = S / w_new;          // map.py:451
        }
        __builtin_amdgcn_wave_barrier();
        if (f == 31) {                                       // after every lane of the group has read obs[s]
            obs[s] = w_old + (float)cnt;
            // (two queues: the flag's block total is kept HERE, behind the previous frame's extract, which zeroes the totals — not by the encoder)
            if (dirty_tot && !dirty[s]) atomicAdd(dirty_tot + (s >> 8), 1);
            dirty[s] = 1;                                    // map.py:452
            dir[0] = 0;
            dir[1] = 0;
            if (hl.list && s < first_new) hl.note(s, slot_lin[s]);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int items = (counters[DIF_C_M] + 31) >> 5;                                                // encoder tiles of this frame
        counters[DIF_C_ITEMS] = items;
        if (fc) {           // this integrate's counters for its extract's snapshot: the next frame's front end rewrites the live words beside that extract
            fc[DIF_FC_SHADOW + 0] = counters[DIF_C_N_OCCUPIED]; fc[DIF_FC_SHADOW + 1] = counters[DIF_C_ALLOC_NEW];
            fc[DIF_FC_SHADOW + 2] = counters[DIF_C_M]; fc[DIF_FC_SHADOW + 3] = counters[DIF_C_C]; fc[DIF_FC_SHADOW + 4] = items;
        }
    }
}

struct FuseArgs {
    const long long* rec; const int* rec_next; int* rec_dir; const int* upd_list; float* latent; float* obs; uint8_t* dirty; int* counters;
    const int64_t* slot_lin; HaloLists hl; dif_pending_export_t* pending;
    int* dirty_tot; int* fc;      // two queues (dif_map_t.frame_seq): block totals of the dirty flags, the frame's counter block
};

__global__ void __launch_bounds__(DIF_BLOCK) k_fuse(FuseArgs a) {
    fuse_body(a.rec, a.rec_next, a.rec_dir, a.upd_list, a.latent, a.obs, a.dirty, a.counters, a.slot_lin, a.hl, a.pending, a.dirty_tot, a.fc);
}
__global__ void __launch_bounds__(DIF_BLOCK) k_fuse_batch(Batch<FuseArgs> b) {
    const FuseArgs& a = b.s[blockIdx.y];
    fuse_body(a.rec, a.rec_next, a.rec_dir, a.upd_list, a.latent, a.obs, a.dirty, a.counters, a.slot_lin, a.hl, a.pending, a.dirty_tot, a.fc);
}
