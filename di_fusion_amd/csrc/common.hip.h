// Shared device helpers for libdifusion (gfx950 / CDNA4 only: wave64, no CUDA compatibility paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/difusion.h"

#define DIF_WAVE 64
#define DIF_BLOCK 256
#define DIF_INVALID_KEY 0xFFFFFFFFu   // slot key of a (point, offset) pair that contributes to no voxel
// No packed fp32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) anywhere in this library.  On MI355X a v_pk_fma_f32 that runs while
// other global loads of its wave are still outstanding (a counted s_waitcnt vmcnt(n > 0) in front of it) now and then loses its write to the LOW
// register of the destination pair in lanes 48..63 once the CU's matrix pipes are kept busy by other waves: tools/micro/pk_fma_fold.hip reproduces
// it on its own (0 wrong folds without MFMA waves, 5-13 in 2-3 M with six, up to 0.1 % with seven; never with v_fma_f32, never behind
// s_waitcnt vmcnt(0); profiles/r06_experiments.md 5).  The SLP vectoriser is off for the whole build (di_fusion_amd/_build.py); the loops the LOOP
// vectoriser would turn into packed pairs carry this pragma, and tests/test_abi.py checks that the built code object holds none.
#define NO_PACKED_F32 _Pragma("clang loop vectorize(disable) interleave(disable)")

#define DIF_CHECK_LAUNCH()                                   \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return DIF_ELAUNCH; \
    } while (0)

namespace dif {

struct Geo {
    int nx, ny, nz;
    float bx, by, bz, vs;
};

__host__ inline Geo geo_of(const dif_map_t* m) {
    Geo g;
    g.nx = m->nx; g.ny = m->ny; g.nz = m->nz;
    g.bx = m->bound_min[0]; g.by = m->bound_min[1]; g.bz = m->bound_min[2];
    g.vs = m->voxel_size;
    return g;
}

// XCD-aware dealing of n work items to the workgroups of a 1-D range: workgroups go to the eight XCDs round-robin by index and every XCD
// has its own L2, so items that follow each other (and share what they read: neighbouring voxels) should sit on ONE XCD.  The n items are
// cut into blocks of 8 R, R the largest power of two <= max_run that fits what is left (190 items, max_run 16: 128 + 32 + 16 + 8, the last 6
// as they come); within a block workgroup 8 j + x takes item x R + j: an XCD gets a run of R consecutive items and every block loads the
// eight XCDs evenly.  A permutation of [0, n) that moves an item at most 8 max_run workgroups away from its own index; b >= n maps to itself.
__device__ __forceinline__ int xcd_run_item(int b, int n, int max_run) {
    if (b >= n) return b;
    for (int base = 0, rem = n; rem >= 16;) {
        int R = 1 << (28 - __clz(rem));                          // 8 R <= rem < 16 R
        if (R > max_run) R = max_run;
        if (b < base + 8 * R) { const int l = b - base; return base + (l & 7) * R + (l >> 3); }
        base += 8 * R; rem -= 8 * R;
    }
    return b;
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// xn = (p - bound_min) / voxel_size : IEEE-754 correctly rounded subtraction and DIVISION (map.py:366-367).
// The library is built with -ffp-contract=off; __fdiv_rn keeps the division exact even if fast-math flags leak in.
__device__ __forceinline__ float normalize1(float p, float b, float vs) { return __fdiv_rn(__fsub_rn(p, b), vs); }

// Voxel (i, i+1] owns xn: id = ceil(xn) - 1 (map.py:368).  Returns false for NaN / out-of-grid.
__device__ __forceinline__ bool voxel_of(const Geo& g, float x, float y, float z, float& xnx, float& xny, float& xnz,
                                         int& ix, int& iy, int& iz) {
    xnx = normalize1(x, g.bx, g.vs);
    xny = normalize1(y, g.by, g.vs);
    xnz = normalize1(z, g.bz, g.vs);
    float cx = ceilf(xnx), cy = ceilf(xny), cz = ceilf(xnz);
    // NaN fails every comparison below
    bool ok = (cx >= 1.0f) && (cy >= 1.0f) && (cz >= 1.0f) && (cx <= (float)g.nx) && (cy <= (float)g.ny) && (cz <= (float)g.nz);
    ix = ok ? (int)cx - 1 : 0;
    iy = ok ? (int)cy - 1 : 0;
    iz = ok ? (int)cz - 1 : 0;
    return ok;
}

__device__ __forceinline__ int linearize(const Geo& g, int ix, int iy, int iz) { return iz + g.nz * iy + (g.nz * g.ny) * ix; }

__device__ __forceinline__ void unlinearize(const Geo& g, int lin, int& ix, int& iy, int& iz) {
    ix = lin / (g.ny * g.nz);
    iy = (lin / g.nz) % g.ny;
    iz = lin % g.nz;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---- grid bitmap whose scan totals are kept up to date by the markers -------------------------------------------------------------
// The ordered scans over the grid bitmap (allocation candidates, voxels to decode) need the number of set bits per scan block.  A
// marker knows when it is the first to set a bit (atomicOr returns the old word), so it counts it into the block's total right away and
// the scan's counting pass — one launch over the whole bitmap per scan — is not needed.  `tot` is idle 0 (zeroed by the kernel that
// follows the scan); `per_words` = words per scan block, the partition of scan_range() for the same grid.
struct GridMarks {
    uint32_t* bits; int* tot; int per_words;
    __device__ __forceinline__ void set(int v) const {
        const uint32_t b = 1u << (v & 31);
        const int w = v >> 5;
        const uint32_t old = atomicOr(bits + w, b);
        if (!(old & b)) atomicAdd(tot + w / per_words, 1);
    }
};

// ---- boundary change lists of a spatially tiled map (dif_map_t.halo_list) ----------------------------------------------------------
// Whoever allocates or fuses an OWNED voxel of the left / right boundary layers appends its slot; dif_export_halo_delta turns the lists
// into the frame's halo messages.  list == nullptr: off (single map, or a caller that only uses whole-layer messages).
struct HaloLists {
    int32_t* list; int cap; int* counters;
    int64_t l_lo, l_hi, r_lo, r_hi;         // linear-id ranges of the two boundary layer sets (empty: no neighbour on that side)
    __device__ __forceinline__ void note(int slot, int64_t lin) const {
        if (!list) return;
        if (lin >= l_lo && lin < l_hi) { const int k = atomicAdd(counters + DIF_C_HALO_L, 1); if (k < cap) list[k] = slot; }
        if (lin >= r_lo && lin < r_hi) { const int k = atomicAdd(counters + DIF_C_HALO_R, 1); if (k < cap) list[cap + k] = slot; }
    }
};

__host__ inline bool map_is_tiled(const dif_map_t* m) { return m->own_x_hi > m->own_x_lo && (m->own_x_lo > 0 || m->own_x_hi < m->nx); }

__host__ inline HaloLists halo_lists_of(const dif_map_t* m) {
    HaloLists h = {};
    if (!m->halo_list || m->halo_list_cap <= 0 || !map_is_tiled(m) || m->halo <= 0) return h;
    const int64_t plane = (int64_t)m->ny * m->nz;
    h.list = m->halo_list; h.cap = m->halo_list_cap; h.counters = m->counters;
    if (m->own_x_lo > 0) { h.l_lo = m->own_x_lo * plane; h.l_hi = (int64_t)(m->own_x_lo + m->halo < m->own_x_hi ? m->own_x_lo + m->halo : m->own_x_hi) * plane; }
    if (m->own_x_hi < m->nx) { h.r_hi = m->own_x_hi * plane; h.r_lo = (int64_t)(m->own_x_hi - m->halo > m->own_x_lo ? m->own_x_hi - m->halo : m->own_x_lo) * plane; }
    return h;
}

// ---- wave / block primitives ---------------------------------------------------------------------------------
// 64-bit value of another lane through DPP (VALU data path; __shfl_up goes through the LDS crossbar: two ds_bpermute per 64-bit value)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ long long dpp_mov64(long long v) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(v >> 32), CTRL, ROW_MASK, 0xF, false);
    return ((long long)hi << 32) | (unsigned int)lo;
}

// Inclusive SEGMENTED scan along the 32 columns of each half-wave: lane (col) adds up its run [my_head .. col].  Four row-local
// steps (row_shr 1, 2, 4, 8 inside the 16-lane DPP rows) and one cross-row step (row_bcast15: lane 15 of the first row to the lanes
// of the second row whose run began in the first).
__device__ __forceinline__ long long seg_incl_scan32(long long v, int col, int my_head) {
    const int head_r = max(my_head, col & 16);                  // where the run starts inside this lane's row
    long long u;
    u = dpp_mov64<0x111, 0xF>(v); if (col - 1 >= head_r) v += u;
    u = dpp_mov64<0x112, 0xF>(v); if (col - 2 >= head_r) v += u;
    u = dpp_mov64<0x114, 0xF>(v); if (col - 4 >= head_r) v += u;
    u = dpp_mov64<0x118, 0xF>(v); if (col - 8 >= head_r) v += u;
    u = dpp_mov64<0x142, 0xA>(v); if (col >= 16 && my_head < 16) v += u;
    return v;
}

__device__ __forceinline__ int wave_incl_scan(int v) {
    int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

// Exclusive scan across a block of up to 1024 threads.  `smem` must hold >= blockDim.x/64 ints.  Returns the exclusive prefix; total in `total`.
__device__ __forceinline__ int block_excl_scan(int v, int* smem, int& total) {
    int lane = lane_id(), wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int inc = wave_incl_scan(v);
    __syncthreads();                       // protect smem from the previous use
    if (lane == 63) smem[wid] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < nw; ++w) {
        int s = smem[w];
        if (w < wid) base += s;
        tot += s;
    }
    total = tot;
    return base + inc - v;
}

__device__ __forceinline__ int block_sum(int v, int* smem) {
    int lane = lane_id(), wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    __syncthreads();
    if (lane == 0) smem[wid] = v;
    __syncthreads();
    int tot = 0;
    for (int w = 0; w < nw; ++w) tot += smem[w];
    return tot;
}

// ---- generic ordered two-pass scan over n elements ------------------------------------------------------------
// F provides:  __device__ int count(int i) const;   __device__ void emit(int i, int offset) const;
//              __device__ void finish(int total) const;   (called once, by block 0 thread 0 of pass 2)
// Element i gets offset = sum_{j<i} count(j).  Output order == index order => deterministic, sorted compaction.
__device__ __forceinline__ void scan_range(int n, int& lo, int& hi) {
    int per = (n + (int)gridDim.x - 1) / (int)gridDim.x;
    per = (per + DIF_BLOCK - 1) / DIF_BLOCK * DIF_BLOCK;
    long long l = (long long)blockIdx.x * per;
    lo = l < n ? (int)l : n;
    long long h = l + per;
    hi = h < n ? (int)h : n;
}

template <class F>
__global__ void __launch_bounds__(DIF_BLOCK) k_scan_pass1(F f, const int* n_ptr, int n_static, int* block_tot) {
    __shared__ int smem[8];
    int n = n_ptr ? *n_ptr : n_static;
    int lo, hi;
    scan_range(n, lo, hi);
    int c = 0;
    for (int i = lo + (int)threadIdx.x; i < hi; i += DIF_BLOCK) c += f.count(i);
    int tot = block_sum(c, smem);
    if (threadIdx.x == 0) block_tot[blockIdx.x] = tot;
}

template <class F>
__global__ void __launch_bounds__(DIF_BLOCK) k_scan_pass2(F f, const int* n_ptr, int n_static, const int* block_tot) {
    __shared__ int smem[8];
    int n = n_ptr ? *n_ptr : n_static;
    int lo, hi;
    scan_range(n, lo, hi);
    int before = 0, all = 0;
    for (int b = (int)threadIdx.x; b < (int)gridDim.x; b += DIF_BLOCK) {
        int t = block_tot[b];
        all += t;
        if (b < (int)blockIdx.x) before += t;
    }
    int offset = block_sum(before, smem);
    int total = block_sum(all, smem);
    for (int base = lo; base < hi; base += DIF_BLOCK) {
        int i = base + (int)threadIdx.x;
        int c = (i < hi) ? f.count(i) : 0;
        int chunk_total;
        int ex = block_excl_scan(c, smem, chunk_total);
        if (i < hi && c > 0) f.emit(i, offset + ex);
        offset += chunk_total;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) f.finish(total);
}

// Single-workgroup variant (one launch instead of two) for small n: thread t owns the contiguous range
// [t*per, (t+1)*per), so offsets still follow index order.
template <class F>
__global__ void __launch_bounds__(1024) k_scan_single(F f, const int* n_ptr, int n_static) {
    __shared__ int smem[16];
    const int n = n_ptr ? *n_ptr : n_static;
    const int per = (n + 1023) / 1024;
    const int lo = min(n, (int)threadIdx.x * per), hi = min(n, lo + per);
    int c = 0;
    for (int i = lo; i < hi; ++i) c += f.count(i);
    int total;
    int offset = block_excl_scan(c, smem, total);
    for (int i = lo; i < hi; ++i) {
        int ci = f.count(i);
        if (ci > 0) f.emit(i, offset);
        offset += ci;
    }
    if (threadIdx.x == 0) f.finish(total);
}

// ---- several independent maps per launch (dif_integrate_frames / dif_extract_streams) ---------------------------------------------------
// The per-map arguments of a kernel travel as an ARRAY in the kernel-argument segment; blockIdx.y (kernels whose workgroups belong to
// one map) or a wave-uniform index (the persistent MLP kernels, which walk the maps' tiles as one concatenated range) selects the entry:
// scalar loads from constant memory, nothing to upload or keep in sync on the device.
template <class A, int NS>
struct BatchN { A s[NS]; };
template <class A>
using Batch = BatchN<A, DIF_MAX_STREAMS>;

// The work items (tiles, voxels) of up to NS maps as ONE range: map j owns [pre[j], pre[j + 1]).  Everything is indexed with compile-time
// constants (select chains), so the tables live in scalar registers.
template <int NS, class T>
struct Ranges {
    T pre[NS + 1], cnt[NS];
    __device__ __forceinline__ T total() const { return pre[NS]; }
    // map `sm`, position `lt` inside it and that map's element count `n` for position x < total() of the concatenated range
    __device__ __forceinline__ void locate(T x, int& sm, T& lt, T& n) const {
        sm = 0; lt = x; n = cnt[0];
#pragma unroll
        for (int j = 1; j < NS; ++j)
            if (x >= pre[j]) { sm = j; lt = x - pre[j]; n = cnt[j]; }
    }
};

// k_scan_pass2 / k_scan_single over S maps in one launch: blockIdx.y = map.  Same partition and order per map as the single launch.
template <class F>
struct ScanBatch { F f[DIF_MAX_STREAMS]; const int* tot[DIF_MAX_STREAMS]; };

template <class F>
__global__ void __launch_bounds__(DIF_BLOCK) k_scan_pass2_batch(ScanBatch<F> b, int n_static) {
    __shared__ int smem[8];
    const F& f = b.f[blockIdx.y];
    const int* __restrict__ block_tot = b.tot[blockIdx.y];
    const int n = n_static;
    int lo, hi;
    scan_range(n, lo, hi);
    int before = 0, all = 0;
    for (int k = (int)threadIdx.x; k < (int)gridDim.x; k += DIF_BLOCK) {
        int t = block_tot[k];
        all += t;
        if (k < (int)blockIdx.x) before += t;
    }
    int offset = block_sum(before, smem);
    int total = block_sum(all, smem);
    for (int base = lo; base < hi; base += DIF_BLOCK) {
        int i = base + (int)threadIdx.x;
        int c = (i < hi) ? f.count(i) : 0;
        int chunk_total;
        int ex = block_excl_scan(c, smem, chunk_total);
        if (i < hi && c > 0) f.emit(i, offset + ex);
        offset += chunk_total;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) f.finish(total);
}

template <class F>
__global__ void __launch_bounds__(1024) k_scan_single_batch(ScanBatch<F> b, int n_static) {
    __shared__ int smem[16];
    const F& f = b.f[blockIdx.y];
    const int n = n_static;
    const int per = (n + 1023) / 1024;
    const int lo = min(n, (int)threadIdx.x * per), hi = min(n, lo + per);
    int c = 0;
    for (int i = lo; i < hi; ++i) c += f.count(i);
    int total;
    int offset = block_excl_scan(c, smem, total);
    for (int i = lo; i < hi; ++i) {
        int ci = f.count(i);
        if (ci > 0) f.emit(i, offset);
        offset += ci;
    }
    if (threadIdx.x == 0) f.finish(total);
}

inline int scan_blocks(int64_t n_upper) {
    int64_t b = (n_upper + DIF_BLOCK - 1) / DIF_BLOCK;      // one 256-element chunk per workgroup while that fits 1024 workgroups: shortest chain
    if (b < 1) b = 1;
    if (b > 1024) b = 1024;
    return (int)b;
}

// scan over a bitmap of `n` words whose per-block totals `tot` were maintained by GridMarks: pass 2 only (or the single-workgroup scan)
inline int counted_scan_blocks(int64_t n) { return n <= 4096 ? 1 : scan_blocks(n); }
inline int counted_scan_per(int64_t n) {
    const int nb = counted_scan_blocks(n);
    int per = (int)((n + nb - 1) / nb);
    return (per + DIF_BLOCK - 1) / DIF_BLOCK * DIF_BLOCK;        // scan_range()'s partition
}
// Pass 2 over a FIXED partition (block b owns elements [b*per, (b+1)*per) whatever the live count is) bounded by a device-side count:
// for per-block totals that were maintained while the flags were set, when the number of live elements is only known on the device.
template <class F>
__global__ void __launch_bounds__(DIF_BLOCK) k_scan_pass2_fixed(F f, const int* __restrict__ n_ptr, int per, const int* __restrict__ block_tot) {
    __shared__ int smem[8];
    const int n = *n_ptr;
    const long long l = (long long)blockIdx.x * per;
    const int lo = l < n ? (int)l : n, hi = (l + per) < n ? (int)(l + per) : n;
    if (lo >= hi && blockIdx.x != 0) return;
    int before = 0, all = 0;
    const int n_blk = blockIdx.x == 0 ? (int)gridDim.x : (int)blockIdx.x;          // block 0 also reports the grand total
    for (int b = (int)threadIdx.x; b < n_blk; b += DIF_BLOCK) {
        const int t = block_tot[b];
        all += t;
        if (b < (int)blockIdx.x) before += t;
    }
    int offset = block_sum(before, smem);
    const int total = block_sum(all, smem);
    for (int base = lo; base < hi; base += DIF_BLOCK) {
        const int i = base + (int)threadIdx.x;
        const int c = (i < hi) ? f.count(i) : 0;
        int chunk_total;
        const int ex = block_excl_scan(c, smem, chunk_total);
        if (i < hi && c > 0) f.emit(i, offset + ex);
        offset += chunk_total;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) f.finish(total);
}

// `n_upper` elements at most, `*n_ptr` live ones; tot[b] = count over elements [b*DIF_BLOCK, (b+1)*DIF_BLOCK): one 256-element chunk per
// workgroup whatever the capacity is (a coarser partition makes a workgroup walk several chunks one after the other, and the emit of
// a chunk is a chain of dependent look-ups)
template <class F>
inline int launch_counted_scan_bounded(F f, const int* n_ptr, int64_t n_upper, const int* tot, hipStream_t s) {
    if (n_upper <= 4096) {
        hipLaunchKernelGGL(k_scan_single<F>, dim3(1), dim3(1024), 0, s, f, n_ptr, 0);
    } else {
        hipLaunchKernelGGL(k_scan_pass2_fixed<F>, dim3((int)((n_upper + DIF_BLOCK - 1) / DIF_BLOCK)), dim3(DIF_BLOCK), 0, s, f, n_ptr, DIF_BLOCK, tot);
    }
    return hipGetLastError() == hipSuccess ? DIF_OK : DIF_ELAUNCH;
}

template <class F>
inline int launch_counted_scan(F f, int n, const int* tot, hipStream_t s) {
    if (n <= 4096) {
        hipLaunchKernelGGL(k_scan_single<F>, dim3(1), dim3(1024), 0, s, f, (const int*)nullptr, n);
    } else {
        hipLaunchKernelGGL(k_scan_pass2<F>, dim3(scan_blocks(n)), dim3(DIF_BLOCK), 0, s, f, (const int*)nullptr, n, tot);
    }
    return hipGetLastError() == hipSuccess ? DIF_OK : DIF_ELAUNCH;
}

// launch_counted_scan for S maps (same n for all of them): one launch, blockIdx.y = map
template <class F>
inline int launch_counted_scan_batch(const ScanBatch<F>& b, int S, int n, hipStream_t s) {
    if (n <= 4096) {
        hipLaunchKernelGGL(k_scan_single_batch<F>, dim3(1, S), dim3(1024), 0, s, b, n);
    } else {
        hipLaunchKernelGGL(k_scan_pass2_batch<F>, dim3(scan_blocks(n), S), dim3(DIF_BLOCK), 0, s, b, n);
    }
    return hipGetLastError() == hipSuccess ? DIF_OK : DIF_ELAUNCH;
}

template <class F>
inline int launch_scan(F f, const int* n_ptr, int n_static, int64_t n_upper, int* block_tmp, hipStream_t s) {
    if (n_upper <= 4096) {   // beyond a few elements per thread the serial single-workgroup scan is slower than two launches
        hipLaunchKernelGGL(k_scan_single<F>, dim3(1), dim3(1024), 0, s, f, n_ptr, n_static);
        return hipGetLastError() == hipSuccess ? DIF_OK : DIF_ELAUNCH;
    }
    int nb = scan_blocks(n_upper);
    hipLaunchKernelGGL(k_scan_pass1<F>, dim3(nb), dim3(DIF_BLOCK), 0, s, f, n_ptr, n_static, block_tmp);
    hipLaunchKernelGGL(k_scan_pass2<F>, dim3(nb), dim3(DIF_BLOCK), 0, s, f, n_ptr, n_static, (const int*)block_tmp);
    return hipGetLastError() == hipSuccess ? DIF_OK : DIF_ELAUNCH;
}

// One launch whatever the static bound: for lists that are short in practice (a frame's dirty voxels) even though their capacity is
// large.  Cost grows with n / 1024 per thread, so only for scans whose n rarely passes a few ten thousand.
template <class F>
inline int launch_scan_one_block(F f, const int* n_ptr, int n_static, hipStream_t s) {
    hipLaunchKernelGGL(k_scan_single<F>, dim3(1), dim3(1024), 0, s, f, n_ptr, n_static);
    return hipGetLastError() == hipSuccess ? DIF_OK : DIF_ELAUNCH;
}

}  // namespace dif
