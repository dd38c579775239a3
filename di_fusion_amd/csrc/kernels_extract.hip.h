// extract_mesh front half: a11 dirty/occupied sets, a12-a14 decoder over sample lattices (MFMA), upsample + refine  (part of libdifusion; included by difusion.hip inside its anonymous namespace)
#pragma once

// =================================================================================================================
// a11 : extract — dirty list, confident neighbourhood, batch ids  (map.py:627-637)
// =================================================================================================================
// set the bitmap bits of the confident voxels among `lin` and its 6 allocated neighbours (map.py:628-631)
__device__ __forceinline__ void mark_confident_nbhd(const Geo& g, int lin, float ignore_th, const int64_t* __restrict__ indexer,
                                                    const float* __restrict__ obs, const GridMarks& marks) {
    const uint32_t* bits = marks.bits;
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
    // three rounds of independent loads (slots, observation counts, bitmap words) instead of seven dependent chains
    int64_t slot[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) slot[c] = indexer[cand[c]];
    float w[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) w[c] = slot[c] >= 0 ? obs[slot[c]] : ignore_th;
    uint32_t word[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) word[c] = (w[c] > ignore_th) ? bits[cand[c] >> 5] : 0xFFFFFFFFu;
    // the seven bitmap updates go out together (one round trip), then the first setters count themselves into their scan blocks
    uint32_t prev[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) {
        const uint32_t b = 1u << (cand[c] & 31);
        prev[c] = (word[c] & b) ? 0xFFFFFFFFu : atomicOr(marks.bits + (cand[c] >> 5), b);
    }
#pragma unroll
    for (int c = 0; c < 7; ++c)
        if (!(prev[c] & (1u << (cand[c] & 31)))) atomicAdd(marks.tot + (cand[c] >> 5) / marks.per_words, 1);
}

// Spatial tiling: dirty HALO voxels (flag copied from their owner by the halo refresh) are not meshed here, but they pull their
// confident neighbourhood into the decoded batch exactly as they do in the single-map run (the blend of a corner depends on
// which neighbours are in the batch, mc_interp_kernel.cu:17-24).
__global__ void __launch_bounds__(DIF_BLOCK) k_mark_halo_dirty(Geo g, float ignore_th, uint8_t* __restrict__ dirty, const int64_t* __restrict__ pos,
                                                             const int64_t* __restrict__ indexer, const float* __restrict__ obs,
                                                             GridMarks bits, const int* __restrict__ n_ptr, int64_t own_lo,
                                                             int64_t own_hi) {
    const int n = *n_ptr;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
        if (!dirty[s]) continue;
        const int64_t p = pos[s];
        if (p >= own_lo && p < own_hi) continue;
        dirty[s] = 0;
        mark_confident_nbhd(g, (int)p, ignore_th, indexer, obs, bits);
    }
}

// Ordered compaction of the dirty set -> valid_blocks (lin ids in ascending SLOT order = sorted `updated_vec_id`, map.py:303-308,627),
// clearing it; each dirty voxel also marks the confident voxels among itself and its 6 allocated neighbours in the grid bitmap
// (map.py:628-631).  The set is a flag per slot.
struct DirtySet {
    uint8_t* dirty;
    const int64_t* pos;
    int64_t* valid_blocks;
    int* counters;
    int no_cache;           // map.py:614-616: every allocated voxel counts as dirty
    int64_t max_voxels;
    Geo g;
    float ignore_th;
    const int64_t* indexer;
    const float* obs;
    GridMarks bits;
    int64_t own_lin_lo, own_lin_hi;     // only owned voxels are meshed (spatial tiling); the whole grid by default
    bool tiled;
    __device__ __forceinline__ bool owned(int s) const {
        if (!tiled) return true;
        const int64_t p = pos[s];
        return p >= own_lin_lo && p < own_lin_hi;               // halo voxels are meshed by their owner
    }
};

// The generic two-pass ordered scan with one element per SLOT: each emit marks its own voxel's neighbourhood, spread over many
// workgroups.  (Measured and rejected: the dirty set as a bitmap over slots compacted word-wise by ONE single-workgroup launch that
// then marks the K neighbourhoods — the ~4,000 grid-bitmap atomics of a frame issued from a single CU serialise: 46 us against
// 12.5 us for the two passes.)
struct DirtyFunctor {
    DirtySet a;
    __device__ int count(int s) const { return ((a.no_cache || a.dirty[s]) && a.owned(s)) ? 1 : 0; }
    __device__ void emit(int s, int offset) const {
        a.dirty[s] = 0;
        if (offset >= a.max_voxels) return;
        const int lin = (int)a.pos[s];
        a.valid_blocks[offset] = lin;
        mark_confident_nbhd(a.g, lin, a.ignore_th, a.indexer, a.obs, a.bits);
    }
    __device__ void finish(int total) const {
        if (total > a.max_voxels) { total = (int)a.max_voxels; a.counters[DIF_C_OVERFLOW] = 2; }
        a.counters[DIF_C_K] = total;
        a.counters[DIF_C_DEFERRED] = 0;
    }
};

// The stream's dirty-set compaction (DirtyFunctor through k_scan_pass2_fixed: block b owns slots [256 b, 256 b + 256), the per-block totals
// were kept by whoever set the flags) with its memory chain started EARLY: a thread reads its flag without waiting for n_occupied (flags
// beyond it are never set), and a dirty slot's position -> 7 neighbour look-ups -> 7 observation counts -> 7 bitmap words are requested
// before the two block-wide sums of the prefix, not after them — nine dependent hops become six.
// DEFERRAL instead of overflow: the per-voxel extract buffers of a stream are sized by the high-water mark of what its frames decoded, not by the
// map's capacity (7.7 KB per row).  A frame decodes B <= min(7 K, n_occupied) voxels (the dirty voxels and their confident 6-neighbours); when that
// bound exceeds the rows there are, every workgroup leaves the launch BEFORE it has changed anything — flags, block totals and bitmap untouched —,
// K = 0 goes to the counters and the bound to counters[DIF_C_DEFERRED]: the frame's extract then finds nothing to do, the caller (which sees the
// word with the frame's snapshot) grows its buffers, and the next extract meshes the accumulated dirty set — the mesh a voxel ends up with is the
// same, one or two frames later (include/difusion.h).
__device__ __forceinline__ void dirty_scan_body(const DirtySet& a, const int* __restrict__ n_ptr, const int* __restrict__ block_tot,
                                                uint32_t* __restrict__ fused_word, int seq) {
    __shared__ int smem[8];
    // two queues: this kernel runs => the frame's fusion kernel, in front of it on this stream, has completed (and released its stores): say so
    // to the other queue, where the next frame's front end waits for exactly that (dif_map_t.frame_seq)
    if (fused_word && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(fused_word, (uint32_t)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int s = (int)blockIdx.x * DIF_BLOCK + (int)threadIdx.x;           // grid covers the capacity (a multiple of DIF_BLOCK)
    const bool flag = a.dirty[s] != 0;
    if (!__syncthreads_or((int)flag) && blockIdx.x != 0) return;            // nothing dirty among this block's 256 slots (most blocks of a frame)
    int before = 0, all = 0;
    const int n_blk = (int)gridDim.x;                                      // every block needs the grand total (deferral), block 0 reports it
    for (int b = (int)threadIdx.x; b < n_blk; b += DIF_BLOCK) {
        const int t = block_tot[b];
        all += t;
        if (b < (int)blockIdx.x) before += t;
    }
    // ---- the dirty voxel's confident neighbourhood: loads now, bitmap updates after the offsets are known (mark_confident_nbhd) ----
    const GridMarks& marks = a.bits;
    int lin = 0, cand[7];
    uint32_t word[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) { cand[c] = 0; word[c] = 0xFFFFFFFFu; }
    // (a stray flag on a slot without a voxel — position -1 beyond n_occupied — is never followed: nothing hangs off an invalid position)
    if (flag && (lin = (int)a.pos[s]) >= 0) {
        int ix, iy, iz;
        unlinearize(a.g, lin, ix, iy, iz);
        cand[0] = lin;
        cand[1] = linearize(a.g, clampi(ix - 1, 0, a.g.nx - 1), iy, iz);
        cand[2] = linearize(a.g, clampi(ix + 1, 0, a.g.nx - 1), iy, iz);
        cand[3] = linearize(a.g, ix, clampi(iy - 1, 0, a.g.ny - 1), iz);
        cand[4] = linearize(a.g, ix, clampi(iy + 1, 0, a.g.ny - 1), iz);
        cand[5] = linearize(a.g, ix, iy, clampi(iz - 1, 0, a.g.nz - 1));
        cand[6] = linearize(a.g, ix, iy, clampi(iz + 1, 0, a.g.nz - 1));
        int64_t slot[7];
#pragma unroll
        for (int c = 0; c < 7; ++c) slot[c] = a.indexer[cand[c]];
        float w[7];
#pragma unroll
        for (int c = 0; c < 7; ++c) w[c] = slot[c] >= 0 ? a.obs[slot[c]] : a.ignore_th;
#pragma unroll
        for (int c = 0; c < 7; ++c) word[c] = (w[c] > a.ignore_th) ? marks.bits[cand[c] >> 5] : 0xFFFFFFFFu;
    }
    const int n = *n_ptr;
    const bool mine = flag && s < n;
    int offset = block_sum(before, smem);
    const int total = block_sum(all, smem);
    {
        const int64_t need = min((int64_t)7 * total, (int64_t)n);          // map.py:628-631: the batch is a subset of the dirty voxels' 7-neighbourhoods
        if (need > a.max_voxels) {
            if (blockIdx.x == 0 && threadIdx.x == 0) { a.counters[DIF_C_K] = 0; a.counters[DIF_C_DEFERRED] = (int)need; }
            return;
        }
    }
    int chunk_total;
    const int ex = block_excl_scan(mine ? 1 : 0, smem, chunk_total);
    if (mine) {
        a.dirty[s] = 0;
        if (offset + ex < a.max_voxels) {
            a.valid_blocks[offset + ex] = lin;
            uint32_t prev[7];
#pragma unroll
            for (int c = 0; c < 7; ++c) {
                const uint32_t b = 1u << (cand[c] & 31);
                prev[c] = (word[c] & b) ? 0xFFFFFFFFu : atomicOr(marks.bits + (cand[c] >> 5), b);
            }
#pragma unroll
            for (int c = 0; c < 7; ++c)
                if (!(prev[c] & (1u << (cand[c] & 31)))) atomicAdd(marks.tot + (cand[c] >> 5) / marks.per_words, 1);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int t = total;
        if (t > a.max_voxels) { t = (int)a.max_voxels; a.counters[DIF_C_OVERFLOW] = 2; }
        a.counters[DIF_C_K] = t;
        a.counters[DIF_C_DEFERRED] = 0;
    }
}

__global__ void __launch_bounds__(DIF_BLOCK) k_dirty_scan(DirtySet a, const int* __restrict__ n_ptr, const int* __restrict__ block_tot, uint32_t* __restrict__ fused_word, int seq) {
    dirty_scan_body(a, n_ptr, block_tot, fused_word, seq);
}
struct DirtyScanArgs { DirtySet a; const int* n_ptr; const int* block_tot; };
__global__ void __launch_bounds__(DIF_BLOCK) k_dirty_scan_batch(Batch<DirtyScanArgs> b) {
    const DirtyScanArgs& a = b.s[blockIdx.y];
    dirty_scan_body(a.a, a.n_ptr, a.block_tot, nullptr, 0);
}
// one wave: a word for another hardware queue's hipStreamWaitValue32 (everything in front of this kernel on its stream has completed)
__global__ void k_publish_word(uint32_t* __restrict__ word, uint32_t value) {
    if (threadIdx.x == 0) __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct OccFunctor {         // bitmap -> occ_slot[b] in ascending lin order; vbm[slot] = b; clears the bitmap
    uint32_t* bits;
    const int64_t* indexer;
    int32_t* occ_slot;
    int32_t* vbm;
    int* counters;
    int64_t max_voxels;
    __device__ int count(int w) const { return __popc(bits[w]); }
    __device__ void emit(int w, int offset) const {
        uint32_t word = bits[w];
        bits[w] = 0u;
        while (word) {
            int b = __ffs((int)word) - 1;
            word &= word - 1;
            int slot = (int)indexer[w * 32 + b];
            if (offset < max_voxels) {
                occ_slot[offset] = slot;
                vbm[slot] = offset;
            }
            ++offset;
        }
    }
    __device__ void finish(int total) const {
        if (total > max_voxels) { total = (int)max_voxels; counters[DIF_C_OVERFLOW] = 3; }
        counters[DIF_C_B] = total;
        counters[DIF_C_VH] = 0;
        counters[DIF_C_WORK] = 0;
    }
};

// =================================================================================================================
// a12..a14 : decoder over the per-voxel sample lattice, fast two-level refinement  (map.py:640-687)
// =================================================================================================================
struct Lattice {            // get_samples(res, a, b) - 0.5 (utility.py:129-149, map.py:645-646): fl(fl(i)*vsize) + a, then - 0.5
    int res;
    float vsize, a;
    __device__ __forceinline__ float coord(int i) const { return ((float)i * vsize + a) - 0.5f; }
};

// decode mode: 0 = lattice (rows are (voxel b, sample s)), 1 = refine list, 2 = explicit rows, 3 = map point query (compacted list of valid points),
// 4 = map point query over ALL points of a posed cloud (dif_sdf_hg): row i = point i, transformed by `pose` and tested for validity here (map.py:565-572);
//     rows that are not valid get out_std = 0 (a valid std is >= 0.05) and nothing else
struct DecodeArgs {
    int mode;
    const int* n_ptr;               // device row / voxel count (modes 0,1,3), or NULL
    int64_t n_static;               // mode 2
    Lattice lat;                    // modes 0,1
    const int32_t* occ_slot;        // modes 0,1 : batch -> slot
    const float* latent;            // modes 0,1,3
    const int32_t* list;            // mode 1: b*R3+sb ; mode 3: point index
    const float* rows;              // mode 2: (n,32)
    const float* xyz;               // mode 3
    const int64_t* indexer;         // mode 3
    Geo geo;                        // mode 3
    float* out_sdf;
    float* out_std;
    float sign;                     // -1 to store the negated sdf (map.py:687)
    float* out_grad;                // GRAD kernels: (n,3) d sdf / d xyz (world units), mode 3 (or d sdf / d x0[29..31] for mode 2)
    const float* wbwd;              // GRAD kernels: transposed-layer blob
    float grad_scale;               // 1 / voxel_size (mode 3), 1 (mode 2)
    const float* fold_table;        // mode 1, optional: [batch voxel][256] constants written by k_decode_voxels (decoder_tile_folded)
    const float* obs;               // mode 4: voxel_obs_count
    float ignore_th;                // mode 4
    float pose[12];                 // mode 4: rows of [R | t] applied to xyz first (x R^T + t, unfused, left to right)
    int* zero_word;                 // optional: a word the launch returns to 0 before anything else (the ticket of the kernel behind it)
};

// x' = x R^T + t the way `other @ th_R.t() + th_t` (utils/motion_util.py:324-327) rounds it in float32: 3-term dot product left to right, then + t
__device__ __forceinline__ float pose_row(const float* T, int j, float x, float y, float z) {
    return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(x, T[4 * j]), __fmul_rn(y, T[4 * j + 1])), __fmul_rn(z, T[4 * j + 2])), T[4 * j + 3]);
}

// Row of a decode tile: which sample lane `col` of tile `tile` works on, where its result goes, and its input fragment
// xin[t] = x0[2t + half] with x0 = [latent 29 | xyz 3] (the natural k order of layer 0 and of the skip input).
struct DecodeRow { bool live; int64_t out_idx; float px, py, pz; };
// DENSE: the instantiation for mode 4 only (and the others without it): the dense query's pose, counts and threshold are twelve more kernel arguments
// held in scalar registers, which the general kernels do not have to spare (k_decode_x6 spilled seven vector registers with both in one body).
template <bool DENSE = false>
__device__ __forceinline__ DecodeRow decode_row_input(const DecodeArgs& A, int64_t tile, int col, int half, int res3, int tiles_per_voxel, int64_t n_rows,
                                                      f16v& xin) {
    DecodeRow R{false, 0, 0.f, 0.f, 0.f};
    const float* lat_row = nullptr;
    const float* row32 = nullptr;
    if (!DENSE && A.mode == 0) {
        const int64_t b = tile / tiles_per_voxel;
        const int s = (int)(tile - b * tiles_per_voxel) * 32 + col;
        R.live = s < res3;
        if (R.live) {
            const int r = A.lat.res;
            R.px = A.lat.coord(s / (r * r)); R.py = A.lat.coord((s / r) % r); R.pz = A.lat.coord(s % r);
            lat_row = A.latent + (int64_t)A.occ_slot[b] * L;
            R.out_idx = b * res3 + s;
        }
    } else if (!DENSE && A.mode == 1) {
        const int64_t row = tile * 32 + col;
        R.live = row < n_rows;
        if (R.live) {
            const int e = A.list[row];
            const int b = e / res3, s = e - b * res3, r = A.lat.res;
            R.px = A.lat.coord(s / (r * r)); R.py = A.lat.coord((s / r) % r); R.pz = A.lat.coord(s % r);
            lat_row = A.latent + (int64_t)A.occ_slot[b] * L;
            R.out_idx = e;
        }
    } else if (!DENSE && A.mode == 2) {
        const int64_t row = tile * 32 + col;
        R.live = row < n_rows;
        if (R.live) { row32 = A.rows + row * 32; R.out_idx = row; }
    } else if (DENSE) {
        const int64_t row = tile * 32 + col;
        if (row < n_rows) {
            const float x = A.xyz[row * 3 + 0], y = A.xyz[row * 3 + 1], z = A.xyz[row * 3 + 2];
            float xn, yn, zn; int ix, iy, iz;
            bool ok = voxel_of(A.geo, pose_row(A.pose, 0, x, y, z), pose_row(A.pose, 1, x, y, z), pose_row(A.pose, 2, x, y, z), xn, yn, zn, ix, iy, iz);
            int64_t slot = -1;
            if (ok) {
                slot = A.indexer[linearize(A.geo, ix, iy, iz)];
                ok = slot >= 0 && A.obs[slot] > A.ignore_th;                                              // map.py:568-572
            }
            R.live = ok;
            R.out_idx = row;
            if (ok) {
                R.px = (xn - (float)ix) - 0.5f; R.py = (yn - (float)iy) - 0.5f; R.pz = (zn - (float)iz) - 0.5f;      // map.py:575
                lat_row = A.latent + slot * L;
            } else if (half == 1) A.out_std[row] = 0.0f;
        }
    } else {
        const int64_t row = tile * 32 + col;
        R.live = row < n_rows;
        if (R.live) {
            const int64_t p = A.list[row];
            float xn, yn, zn; int ix, iy, iz;
            voxel_of(A.geo, A.xyz[p * 3 + 0], A.xyz[p * 3 + 1], A.xyz[p * 3 + 2], xn, yn, zn, ix, iy, iz);
            R.px = (xn - (float)ix) - 0.5f; R.py = (yn - (float)iy) - 0.5f; R.pz = (zn - (float)iz) - 0.5f;      // map.py:575
            lat_row = A.latent + A.indexer[linearize(A.geo, ix, iy, iz)] * L;
            R.out_idx = row;
        }
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int k = 2 * t + half;
        float v = 0.0f;
        if (R.live) {
            if (row32) v = row32[k];
            else if (k < L) v = lat_row[k];
            else v = (k == L) ? R.px : ((k == L + 1) ? R.py : R.pz);
        }
        xin[t] = v;
    }
    return R;
}

// GRAD: 256 threads = one wave per SIMD with the full 512-register budget (forward + reverse chain keep ~300 values live)
template <bool GRAD, bool DENSE = false>
__global__ void __launch_bounds__(GRAD ? 256 : 512, GRAD ? 1 : 2) k_decode(DecodeArgs A, const float* __restrict__ wblob) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (DENSE && A.zero_word && blockIdx.x == 0 && threadIdx.x == 0) *A.zero_word = 0;
    stage_weights(lds, wblob, DEC_LDS_FLOATS);
    const __amdgpu_buffer_rsrc_t wfwd = make_rsrc(wblob, DEC_FLOATS);
    const __amdgpu_buffer_rsrc_t wbwd = make_rsrc(GRAD ? A.wbwd : wblob, GRAD ? DECB_FLOATS : DEC_FLOATS);
    const int lane = lane_id(), half = lane >> 5, col = lane & 31;
    // work item w goes to wave (w / #blocks) of block (w % #blocks): a partly filled launch spreads over all CUs and SIMDs first
    const int wave = (int)((threadIdx.x >> 6) * gridDim.x + blockIdx.x);
    const int nwaves = (int)(gridDim.x * (blockDim.x >> 6));
    const int res3 = A.lat.res * A.lat.res * A.lat.res;
    const int tiles_per_voxel = (res3 + 31) / 32;
    int64_t n_rows, n_tiles;
    if (A.mode == 0) {
        n_rows = (int64_t)(*A.n_ptr) * res3;
        n_tiles = (int64_t)(*A.n_ptr) * tiles_per_voxel;
    } else {
        n_rows = A.n_ptr ? (int64_t)(*A.n_ptr) : A.n_static;
        n_tiles = (n_rows + 31) / 32;
    }
    for (int64_t tile = wave; tile < n_tiles; tile += nwaves) {
        f16v xin;
        const DecodeRow R = decode_row_input<DENSE>(A, tile, col, half, res3, tiles_per_voxel, n_rows, xin);
        const bool live = R.live;
        const int64_t out_idx = R.out_idx;
        const float px = R.px, py = R.py, pz = R.pz;
        float sdf, sd;
        if (!GRAD && A.mode == 1 && A.fold_table) {          // refine rows: the voxel's latent terms come ready-made from the lattice pass
            const int b = live ? A.list[tile * 32 + col] / res3 : 0;
            decoder_tile_folded(lds, wfwd, FoldInitGlobal{A.fold_table + (int64_t)b * 256}, px, py, pz, lane, sdf, sd);
        } else if (GRAD) {
            float gx, gy, gz;
            decoder_tile_grad(lds, wfwd, wbwd, xin, lane, sdf, sd, gx, gy, gz);
            if (live && half == 1) {
                A.out_grad[out_idx * 3 + 0] = gx * A.grad_scale;      // d rel / d xyz = 1 / voxel_size (map.py:565,575)
                A.out_grad[out_idx * 3 + 1] = gy * A.grad_scale;
                A.out_grad[out_idx * 3 + 2] = gz * A.grad_scale;
            }
        } else {
            decoder_tile(lds, wfwd, xin, lane, sdf, sd);
        }
        if (live) {
            if (half == 0) A.out_sdf[out_idx] = A.sign * sdf;
            else A.out_std[out_idx] = sd;
        }
    }
}

// Modes 0 (lattice, every sample), 2 (explicit rows) and 3 (map point query, values only) on the bf16 matrix pipe: the row set-up of
// k_decode, the tile of decoder_tile_x6.  wblob = packing.py:pack_decoder_x6, wu = pack_decoder_x6u.
template <bool DENSE = false>
__global__ void __launch_bounds__(512, 1) k_decode_x6(DecodeArgs A, const float* __restrict__ wblob, const float* __restrict__ wu) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (DENSE && A.zero_word && blockIdx.x == 0 && threadIdx.x == 0) *A.zero_word = 0;
    stage_weights(lds, wblob, X6_LDS_BYTES / 4);
    const __amdgpu_buffer_rsrc_t wfwd = make_rsrc(wblob, X6_BYTES / 4);
    const __amdgpu_buffer_rsrc_t wun = make_rsrc(wu, X6U_BYTES / 4);
    const int lane = lane_id(), half = lane >> 5, col = lane & 31;
    const int wave = (int)((threadIdx.x >> 6) * gridDim.x + blockIdx.x);
    const int nwaves = (int)(gridDim.x * (blockDim.x >> 6));
    const int res3 = A.lat.res * A.lat.res * A.lat.res;
    const int tiles_per_voxel = (res3 + 31) / 32;
    int64_t n_rows, n_tiles;
    if (A.mode == 0) {
        n_rows = (int64_t)(*A.n_ptr) * res3;
        n_tiles = (int64_t)(*A.n_ptr) * tiles_per_voxel;
    } else {
        n_rows = A.n_ptr ? (int64_t)(*A.n_ptr) : A.n_static;
        n_tiles = (n_rows + 31) / 32;
    }
    for (int64_t tile = wave; tile < n_tiles; tile += nwaves) {
        f16v xin;
        const DecodeRow R = decode_row_input<DENSE>(A, tile, col, half, res3, tiles_per_voxel, n_rows, xin);
        const bool live = R.live;
        const int64_t out_idx = R.out_idx;
        float sdf, sd;
        decoder_tile_x6(lds, wfwd, wun, xin, lane, sdf, sd);
        if (live) {
            if (half == 0) A.out_sdf[out_idx] = A.sign * sdf;
            else A.out_std[out_idx] = sd;
        }
    }
}

#ifndef GRAD_X6_THREADS
#define GRAD_X6_THREADS 256
#endif
// Values AND d sdf / d xyz (modes 2, 3) on the bf16 matrix pipe: decoder_tile_grad_x6.  One wave per SIMD with the 512-register budget,
// like k_decode<true>; wb = packing.py:pack_decoder_x6_backward.
template <int PF, bool DENSE = false>
__global__ void __launch_bounds__(GRAD_X6_THREADS, 1) k_decode_grad_x6(DecodeArgs A, const float* __restrict__ wblob, const float* __restrict__ wu,
                                                          const float* __restrict__ wb) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (DENSE && A.zero_word && blockIdx.x == 0 && threadIdx.x == 0) *A.zero_word = 0;
    stage_weights(lds, wblob, X6_LDS_BYTES / 4);
    const __amdgpu_buffer_rsrc_t wfwd = make_rsrc(wblob, X6_BYTES / 4);
    const __amdgpu_buffer_rsrc_t wun = make_rsrc(wu, X6U_BYTES / 4);
    const __amdgpu_buffer_rsrc_t wbw = make_rsrc(wb, X6B_BYTES / 4);
    const int lane = lane_id(), half = lane >> 5, col = lane & 31;
    const int wave = (int)((threadIdx.x >> 6) * gridDim.x + blockIdx.x);
    const int nwaves = (int)(gridDim.x * (blockDim.x >> 6));
    const int res3 = A.lat.res * A.lat.res * A.lat.res;
    const int64_t n_rows = A.n_ptr ? (int64_t)(*A.n_ptr) : A.n_static;
    const int64_t n_tiles = (n_rows + 31) / 32;
    for (int64_t tile = wave; tile < n_tiles; tile += nwaves) {
        f16v xin;
        const DecodeRow R = decode_row_input<DENSE>(A, tile, col, half, res3, 1, n_rows, xin);
        float sdf, sd, gx, gy, gz;
        decoder_tile_grad_x6<PF>(lds, wfwd, wun, wbw, xin, lane, sdf, sd, gx, gy, gz);
        if (R.live) {
            if (half == 0) A.out_sdf[R.out_idx] = A.sign * sdf;
            else {
                A.out_std[R.out_idx] = sd;
                A.out_grad[R.out_idx * 3 + 0] = gx * A.grad_scale;      // d rel / d xyz = 1 / voxel_size (map.py:565,575)
                A.out_grad[R.out_idx * 3 + 1] = gy * A.grad_scale;
                A.out_grad[R.out_idx * 3 + 2] = gz * A.grad_scale;
            }
        }
    }
}

// Refine rows (mode 1 with the lattice pass's fold table) on the bf16 matrix pipe; wblob = packing.py:pack_decoder_x6.
// NS > 1: the refine lists of S <= NS maps walked as one range of tiles (see encode_body); lattice and sign are those of map 0.
template <int NS>
__device__ __forceinline__ void decode_refine_x6_body(const BatchN<DecodeArgs, NS>& B, int S, const float* __restrict__ wblob) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = lane_id(), half = lane >> 5, col = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane((int)((threadIdx.x >> 6) * gridDim.x + blockIdx.x));
    const int nwaves = (int)(gridDim.x * (blockDim.x >> 6));
    const Lattice lat = B.s[0].lat;
    const float sign = B.s[0].sign;
    const int res3 = lat.res * lat.res * lat.res, r = lat.res;
    Ranges<NS, int64_t> rg;              // cnt: refine rows of map j; pre: its first tile in the concatenated range
    rg.pre[0] = 0;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        rg.cnt[j] = (j < S) ? (B.s[j].n_ptr ? (int64_t)(*B.s[j].n_ptr) : B.s[j].n_static) : 0;
        rg.pre[j + 1] = rg.pre[j] + (rg.cnt[j] + 31) / 32;
    }
    const int64_t n_tiles = rg.total();
    // the first tile's list entries are requested before the weights are staged (one dependent hop off the critical path)
    int sm_n; int64_t lt_n, rows_n;
    rg.locate(wave, sm_n, lt_n, rows_n);
    int e_next = (wave < n_tiles && lt_n * 32 + col < rows_n) ? B.s[sm_n].list[lt_n * 32 + col] : 0;
    __builtin_amdgcn_sched_barrier(0);
    stage_weights(lds, wblob, X6_LDS_BYTES / 4);
    const __amdgpu_buffer_rsrc_t wfwd = make_rsrc(wblob, X6_BYTES / 4);
    for (int64_t T = wave; T < n_tiles; T += nwaves) {
        const int sm = sm_n;
        const int64_t tile = lt_n, n_rows = rows_n;
        const DecodeArgs& A = B.s[sm];
        const int64_t row = tile * 32 + col;
        const bool live = row < n_rows;
        const int e = e_next;
        rg.locate(T + nwaves, sm_n, lt_n, rows_n);
        e_next = (T + nwaves < n_tiles && lt_n * 32 + col < rows_n) ? B.s[sm_n].list[lt_n * 32 + col] : 0;
        const int b = e / res3, sb = e - b * res3;
        const float px = lat.coord(sb / (r * r)), py = lat.coord((sb / r) % r), pz = lat.coord(sb % r);
        float sdf, sd;
        decoder_tile_folded_x6(lds, wfwd, FoldInitGlobal{A.fold_table + (int64_t)b * 256}, px, py, pz, lane, sdf, sd);
        if (live) {
            if (half == 0) A.out_sdf[e] = sign * sdf;
            else A.out_std[e] = sd;
        }
    }
}

__global__ void __launch_bounds__(512, 1) k_decode_refine_x6(DecodeArgs A, const float* __restrict__ wblob) {
    const BatchN<DecodeArgs, 1> B{{A}};
    decode_refine_x6_body<1>(B, 1, wblob);
}
__global__ void __launch_bounds__(512, 1) k_decode_refine_x6_batch(Batch<DecodeArgs> B, int S, const float* __restrict__ wblob) {
    decode_refine_x6_body<DIF_MAX_STREAMS>(B, S, wblob);
}

// Trilinear x2 upsample (align_corners) of the low lattice + selection of samples to re-decode (map.py:655-667).
// ATen CPU semantics (see oracle.trilinear_upsample_align_corners): per axis src = scale*j, i0 = int(src),
// lam1 = src - i0, lam0 = 1 - lam1, two-tap value = fma(t0, lam0, t1*lam1), w innermost then h then d.
__device__ __forceinline__ void tri_axis(int j, int l, float scale, int& i0, int& i1, float& w0, float& w1) {
    float src = scale * (float)j;
    i0 = min((int)src, l - 1);
    i1 = i0 + ((i0 < l - 1) ? 1 : 0);
    w1 = fminf(fmaxf(src - (float)i0, 0.0f), 1.0f);
    w0 = 1.0f - w1;
}

__device__ __forceinline__ float tri_sample(const float* __restrict__ low, int l, int x0, int x1, int y0, int y1, int z0, int z1,
                                            float wx0, float wx1, float wy0, float wy1, float wz0, float wz1) {
    // layout [x][y][z], z innermost ("w"), x outermost ("d")
    float v00 = fmaf(low[(x0 * l + y0) * l + z0], wz0, low[(x0 * l + y0) * l + z1] * wz1);
    float v01 = fmaf(low[(x0 * l + y1) * l + z0], wz0, low[(x0 * l + y1) * l + z1] * wz1);
    float v10 = fmaf(low[(x1 * l + y0) * l + z0], wz0, low[(x1 * l + y0) * l + z1] * wz1);
    float v11 = fmaf(low[(x1 * l + y1) * l + z0], wz0, low[(x1 * l + y1) * l + z1] * wz1);
    float v0 = fmaf(v00, wy0, v01 * wy1);
    float v1 = fmaf(v10, wy0, v11 * wy1);
    return fmaf(v0, wx0, v1 * wx1);
}

// One thread per (voxel, x, y) row of R samples along z; selected samples are appended to the refine list with ONE atomic per
// workgroup (a per-wave atomic on a single counter costs ~12 ns each and serialises: 15k waves = 200 us).
__global__ void __launch_bounds__(DIF_BLOCK) k_upsample_mark(const float* __restrict__ low_sdf, const float* __restrict__ low_std, int l, int R,
                                                           float* __restrict__ cube_sdf, float* __restrict__ cube_std,
                                                           int32_t* __restrict__ refine_list, int* __restrict__ counters) {
    __shared__ int smem[8];
    __shared__ int s_base;
    const int B = counters[DIF_C_B];
    const int R2 = R * R, R3 = R2 * R, l3 = l * l * l;
    const int64_t n_rows = (int64_t)B * R2;
    const float scale = (float)(l - 1) / (float)(R - 1);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n_pad = (n_rows + DIF_BLOCK - 1) / DIF_BLOCK * DIF_BLOCK;
    for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < n_pad; row += stride) {
        unsigned sel = 0;
        int64_t e0 = 0;
        if (row < n_rows) {
            const int b = (int)(row / R2), jxy = (int)(row - (int64_t)b * R2);
            const int jx = jxy / R, jy = jxy % R;
            int x0, x1, y0, y1; float wx0, wx1, wy0, wy1;
            tri_axis(jx, l, scale, x0, x1, wx0, wx1);
            tri_axis(jy, l, scale, y0, y1, wy0, wy1);
            const float* ls = low_sdf + (int64_t)b * l3;
            const float* ld = low_std + (int64_t)b * l3;
            e0 = (int64_t)b * R3 + (int64_t)jxy * R;
            // (not vectorised: two iterations at a time come out as v_pk_fma_f32 / v_pk_mul_f32 on freshly loaded pairs — common.hip.h:NO_PACKED_F32)
            NO_PACKED_F32
            for (int jz = 0; jz < R; ++jz) {
                int z0, z1; float wz0, wz1;
                tri_axis(jz, l, scale, z0, z1, wz0, wz1);
                float sv = tri_sample(ls, l, x0, x1, y0, y1, z0, z1, wx0, wx1, wy0, wy1, wz0, wz1);
                float dv = tri_sample(ld, l, x0, x1, y0, y1, z0, z1, wx0, wx1, wy0, wy1, wz0, wz1);
                cube_sdf[e0 + jz] = -sv;
                cube_std[e0 + jz] = dv;
                if (fabsf(sv) < 0.05f) sel |= 1u << jz;               // map.py:667
            }
        }
        int total;
        int ex = block_excl_scan(__popc(sel), smem, total);
        if (total > 0) {
            if (threadIdx.x == 0) s_base = atomicAdd(counters + DIF_C_VH, total);
            __syncthreads();
            int o = s_base + ex;
            while (sel) {
                int jz = __ffs((int)sel) - 1;
                sel &= sel - 1;
                refine_list[o++] = (int32_t)(e0 + jz);
            }
        }
        __syncthreads();
    }
}

// Fused low-lattice decode + upsample for the fast two-level scheme (resolution <= 4, i.e. R^2 <= 64 rows = one per lane):
// TWO waves own one voxel — each puts one 32-sample tile of the l^3 low samples through the MLP (l = 4: exactly two tiles) into the
// pair's LDS record, then each does half of the ATen-exact trilinear x2 upsample out of LDS (lane = (x, y) row, the pair splits z),
// writes its half of the cube and appends its |sdf| < 0.05 samples to the global refine list (space reserved once per workgroup and
// round).  A frame decodes ~900 voxels: with one wave per voxel every SIMD ran a single wave through two tiles back to back and the
// matrix pipe idled in that wave's gaps (bias loads, ReLU, heads: ~75 % issue rate); two co-resident waves fill each other's gaps.
// Work per voxel is uniform, so the launch is balanced; the exact re-decode of the selected samples stays a separate, globally
// balanced launch (per-voxel counts range 0..R^3).
struct VoxelDecodeArgs {
    const int32_t* occ_slot;
    const float* latent;
    Lattice low;
    int R;
    float* cube_sdf;
    float* cube_std;
    int32_t* refine_list;
    int* counters;
    const float* fold_w;            // packing.py:pack_decoder_fold, or NULL (latent carried through the MFMAs)
    float* fold_table;              // [batch voxel][256] out, for the refine pass
};

#define VD_MAX_L3 64
#define VD_MAX_R2 64
#define VD_WAVE_LDS_FLOATS (2 * VD_MAX_L3 + 2 * 256) /* per PAIR of waves: low sdf + low std + each wave's copy of the voxel's folded decoder constants */

#ifdef DIF_TRACE            // tools/trace_decode.py: per-wave phase timestamps (100 MHz wall clock) of the last k_decode_voxels launch
__device__ unsigned long long g_vd_trace[2048 * 8];
#define VD_STAMP(slot) do { if (lane_id() == 0) g_vd_trace[((threadIdx.x >> 6) * gridDim.x + blockIdx.x) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define VD_STAMP(slot) do { } while (0)
#endif

// X6: the tiles run on the bf16 matrix pipe (decoder_tile_folded_x6; wblob = packing.py:pack_decoder_x6, folding required).
// NS > 1: the decoded batches of S <= NS maps are walked as ONE range of voxels (map 0's, then map 1's, ...): weights staged once per
// workgroup, S frames' worth of pairs per launch.  A pair's voxel belongs to one map (scalar loads of that map's pointers from the
// kernel-argument array); lattice, resolution and fold weights are those of map 0; refine-list space is reserved per map.
template <bool X6, int NS>
__device__ __forceinline__ void decode_voxels_body(const BatchN<VoxelDecodeArgs, NS>& AB, int S, const float* __restrict__ wblob) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    VD_STAMP(0);
    constexpr int LDS_W = X6 ? X6_LDS_BYTES / 4 : ((DEC_LDS_FLOATS + 3) & ~3);
#if defined(DIF_VD_CUT) && DIF_VD_CUT == 0
    return;                                      // (measurement builds only, tools/sweep_decode.py: what an empty launch of this shape costs)
#endif
    stage_weights(lds, wblob, X6 ? X6_LDS_BYTES / 4 : DEC_LDS_FLOATS);
    VD_STAMP(1);
#if defined(DIF_VD_CUT) && DIF_VD_CUT == 1
    return;                                      // ... + the weight staging
#endif
    const __amdgpu_buffer_rsrc_t wfwd = make_rsrc(wblob, X6 ? X6_BYTES / 4 : DEC_FLOATS);
    const int lane = lane_id(), half = lane >> 5, col = lane & 31, wid = threadIdx.x >> 6;
    const int pair = wid >> 1, tsel = wid & 1;                 // wave `tsel` of the pair that owns the voxel
    float* w_low_sdf = lds + LDS_W + pair * VD_WAVE_LDS_FLOATS;       // shared by the pair
    float* w_low_std = w_low_sdf + VD_MAX_L3;
    float* w_fold = w_low_std + VD_MAX_L3 + tsel * 256;        // [c0 | c3], see decoder_fold_consts: each wave keeps its own copy (no barrier before the MFMAs)
    const Lattice low = AB.s[0].low;
    const float* const fold_w = AB.s[0].fold_w;
    const int l = low.res, R = AB.s[0].R, l3 = l * l * l, R2 = R * R, R3 = R2 * R;
    const float scale = (float)(l - 1) / (float)(R - 1);
    int pre[NS + 1];                                           // first voxel of map j in the concatenated range
    pre[0] = 0;
#pragma unroll
    for (int j = 0; j < NS; ++j) pre[j + 1] = pre[j] + ((j < S) ? AB.s[j].counters[DIF_C_B] : 0);
    const int B = pre[NS];
    const int pairs_per_block = (int)(blockDim.x >> 7), n_pairs = (int)gridDim.x * pairs_per_block;
    const int first = __builtin_amdgcn_readfirstlane(pair * (int)gridDim.x + (int)blockIdx.x);  // spread over CUs first
    // uniform trip count: the barriers below are workgroup-wide (the pair's hand-over, and the refine-list reservation: one global
    // atomic per workgroup, map and round instead of one per voxel — 900 same-address atomics at the end of the launch queued up for ~5 us)
    __shared__ int s_tot[8], s_map[8], s_base[NS];
    const int rounds = (B + n_pairs - 1) / n_pairs;
    for (int round = 0; round < rounds; ++round) {
        const int bg = first + round * n_pairs;                // index in the concatenated range
        int sm = 0, b = bg;                                    // map and index in that map's batch
#pragma unroll
        for (int j = 1; j < NS; ++j)
            if (bg >= pre[j]) { sm = j; b = bg - pre[j]; }
        const VoxelDecodeArgs& A = AB.s[sm];
        unsigned sel = 0;
        const int jz0 = tsel * (R >> 1), jz1 = jz0 + (R >> 1);  // this wave's share of the z samples
        const int64_t e0 = (int64_t)b * R3 + (int64_t)lane * R;
        if (bg < B && tsel * 32 < l3) {
            const float* lat_row = A.latent + (int64_t)A.occ_slot[b] * L;
            const int s = tsel * 32 + col;
            const float px = low.coord(s / (l * l)), py = low.coord((s / l) % l), pz = low.coord(s % l);
            float sdf = 0.0f, sd = 0.0f;
            // ---- this wave's tile of the low lattice -> LDS (map.py:644-653) ----
            if (X6 || fold_w) {
                // the voxel's latent goes through lin0 / lin3 once (VALU), every sample then only adds its coordinate columns (MFMA)
                if constexpr (X6) decoder_fold_consts_x6(lds, fold_w, lat_row, w_fold, lane);
                else decoder_fold_consts(lds, fold_w, lat_row, w_fold, lane);
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xc07f);
                if (tsel == 0) {
                    float* rec = A.fold_table + (int64_t)b * 256;          // the refine pass picks the constants up from here
                    for (int p = lane; p < 256; p += 64) rec[p] = w_fold[p];
                }
                VD_STAMP(6);
#if defined(DIF_VD_CUT) && DIF_VD_CUT == 2
                sdf = w_fold[lane]; sd = w_fold[lane + 64];      // ... + the fold constants, no MLP tile (the rest of the round runs on garbage)
#else
                if constexpr (X6) decoder_tile_folded_x6(lds, wfwd, FoldInitLds{w_fold}, px, py, pz, lane, sdf, sd);
                else decoder_tile_folded(lds, wfwd, FoldInitLds{w_fold}, px, py, pz, lane, sdf, sd);
#endif
            } else if constexpr (!X6) {
                f16v xin;
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int k = 2 * t + half;
                    xin[t] = (k < L) ? lat_row[k] : 0.0f;
                }
                if (half) { xin[14] = px; xin[15] = pz; } else { xin[15] = py; }        // k = 29 (x), 30 (y), 31 (z)
                decoder_tile(lds, wfwd, xin, lane, sdf, sd);
            }
            if (s < l3) {
                if (half == 0) w_low_sdf[s] = sdf;
                else w_low_std[s] = sd;
            }
        }
        __syncthreads();                        // both tiles of every pair are in LDS
        VD_STAMP(2);
        // ---- trilinear x2 + threshold (map.py:655-667): lane = (jx, jy) row of R samples along z, this wave's half of them ----
        if (bg < B && lane < R2) {
            const int jx = lane / R, jy = lane % R;
            int x0, x1, y0, y1; float wx0, wx1, wy0, wy1;
            tri_axis(jx, l, scale, x0, x1, wx0, wx1);
            tri_axis(jy, l, scale, y0, y1, wy0, wy1);
            float* const cube_sdf = A.cube_sdf;
            float* const cube_std = A.cube_std;
            for (int jz = jz0; jz < jz1; ++jz) {
                int z0, z1; float wz0, wz1;
                tri_axis(jz, l, scale, z0, z1, wz0, wz1);
                float sv = tri_sample(w_low_sdf, l, x0, x1, y0, y1, z0, z1, wx0, wx1, wy0, wy1, wz0, wz1);
                float dv = tri_sample(w_low_std, l, x0, x1, y0, y1, z0, z1, wx0, wx1, wy0, wy1, wz0, wz1);
                cube_sdf[e0 + jz] = -sv;
                cube_std[e0 + jz] = dv;
                if (fabsf(sv) < 0.05f) sel |= 1u << jz;
            }
        }
        VD_STAMP(3);
        const int c = __popc(sel);
        const int incl = wave_incl_scan(c);
        const int total = __shfl(incl, 63);
        if (lane == 0) { s_tot[wid] = total; if (NS > 1) s_map[wid] = sm; }
        __syncthreads();
        if (NS == 1) {
            if (threadIdx.x == 0) {
                int sum = 0;
                for (int w = 0; w < (int)(blockDim.x >> 6); ++w) sum += s_tot[w];
                s_base[0] = sum ? atomicAdd(A.counters + DIF_C_VH, sum) : 0;
            }
        } else if ((int)threadIdx.x < S) {      // thread j reserves for map j what this workgroup's waves selected in map j's voxels
            const int j = (int)threadIdx.x;
            int sum = 0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) sum += (s_map[w] == j) ? s_tot[w] : 0;
            s_base[j] = sum ? atomicAdd(AB.s[j].counters + DIF_C_VH, sum) : 0;
        }
        __syncthreads();
        if (total > 0) {
            int o = s_base[NS == 1 ? 0 : sm] + incl - c;
            for (int w = 0; w < wid; ++w) o += (NS == 1 || s_map[w] == sm) ? s_tot[w] : 0;
            int32_t* const refine_list = A.refine_list;
            while (sel) {
                const int jz = __ffs((int)sel) - 1;
                sel &= sel - 1;
                refine_list[o++] = (int32_t)(e0 + jz);
            }
        }
        __syncthreads();                        // s_tot / s_base and the pair's LDS record are rewritten by the next round
        VD_STAMP(4);
    }
    VD_STAMP(5);
}

template <bool X6>
__global__ void __launch_bounds__(512, X6 ? 1 : 2) k_decode_voxels(VoxelDecodeArgs A, const float* __restrict__ wblob) {
    const BatchN<VoxelDecodeArgs, 1> B{{A}};
    decode_voxels_body<X6, 1>(B, 1, wblob);
}
template <bool X6>
__global__ void __launch_bounds__(512, X6 ? 1 : 2) k_decode_voxels_batch(Batch<VoxelDecodeArgs> B, int S, const float* __restrict__ wblob) {
    decode_voxels_body<X6, DIF_MAX_STREAMS>(B, S, wblob);
}
