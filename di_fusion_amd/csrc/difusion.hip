// libdifusion — MI355X (gfx950) kernels + C ABI for DI-Fusion's per-frame fusion path.  See include/difusion.h.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC difusion.hip -o libdifusion.so
//
// Every float expression whose rounding is observable in voxel ids or lattice coordinates is written with the
// reference's operation order and compiled without FMA contraction; fmaf() is used only where the reference's own
// CPU build fuses (trilinear upsample) or where the order is ours to choose (MLP accumulation = the MFMA's fmaf chain).
#include <cstring>
#include <vector>
#include <hip/hip_runtime.h>

#include "common.hip.h"
#include "mlp.hip.h"
#include "mc_tables.inc"

using namespace dif;

namespace {

constexpr int L = DIF_LATENT_DIM;     // 29
constexpr int ITEM_ROWS = 32;         // gathered rows per encoder work item = one MFMA tile of 32 points (finest load balance)

__device__ __constant__ int c_mc_edge_table[256];
__device__ __constant__ signed char c_mc_tri_table[256][16];
bool g_tables_uploaded[64] = {};

int upload_tables() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return DIF_ELAUNCH;
    if (dev < 64 && g_tables_uploaded[dev]) return DIF_OK;
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_mc_edge_table), k_mc_edge_table, sizeof(k_mc_edge_table)) != hipSuccess) return DIF_ELAUNCH;
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_mc_tri_table), k_mc_tri_table, sizeof(k_mc_tri_table)) != hipSuccess) return DIF_ELAUNCH;
    if (dev < 64) g_tables_uploaded[dev] = true;
    return DIF_OK;
}

int num_cus() {
    static int cus[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 64 && cus[dev]) return cus[dev];
    hipDeviceProp_t p;
    int n = 256;
    if (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) n = p.multiProcessorCount;
    if (dev < 64) cus[dev] = n;
    return n;
}

// ---- optional per-kernel timing (dif_profile_*) -----------------------------------------------------------------
struct ProfRec { hipEvent_t a, b; int which; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
struct ProfScope {
    hipStream_t s; int which; hipEvent_t a = nullptr, b = nullptr;
    ProfScope(int which_, hipStream_t s_) : s(s_), which(which_) {
        if (!g_prof_on || g_prof.size() >= 65536) return;
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;      // never put events into a captured graph
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
        (void)hipEventRecord(a, s);
    }
    ~ProfScope() {
        if (!a) return;
        (void)hipEventRecord(b, s);
        g_prof.push_back(ProfRec{a, b, which});
    }
};

inline int grid_for(int64_t n, int per_block = DIF_BLOCK, int max_blocks = 4096) {
    int64_t b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (int)b;
}

// =================================================================================================================
// a1 / a2 : depth -> points  (ext/imgproc/imgproc.cu:5-44; utils/motion_util.py:322-327)
// =================================================================================================================
// One thread per pixel, threadIdx.x walks u (columns) => coalesced 4 B reads / 12 B writes (the reference walks rows).
__global__ void __launch_bounds__(DIF_BLOCK) k_unproject(const float* __restrict__ depth, float* __restrict__ pc, int H, int W,
                                                       float fx, float fy, float cx, float cy) {
    int64_t n = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int v = (int)(i / W), u = (int)(i - (int64_t)v * W);
        float d = depth[i];
        float x, y, z;
        if (d == d) {
            x = ((float)u - cx) / fx * d;       // (u - cx) / fx * d, imgproc.cu:18
            y = ((float)v - cy) / fy * d;
            z = d;
        } else {
            x = y = z = __builtin_nanf("");
        }
        pc[i * 3 + 0] = x; pc[i * 3 + 1] = y; pc[i * 3 + 2] = z;
    }
}

struct Pose { float r[9]; float t[3]; };

__global__ void __launch_bounds__(DIF_BLOCK) k_unproject_transform(const float* __restrict__ depth, const float* __restrict__ ncam,
                                                                 float* __restrict__ xyz, float* __restrict__ nrm, int H, int W,
                                                                 float fx, float fy, float cx, float cy, Pose P, const float* __restrict__ pose_dev) {
    if (pose_dev) {                                   // pose read from device memory: lets a captured hipGraph be replayed per frame
#pragma unroll
        for (int i = 0; i < 9; ++i) P.r[i] = pose_dev[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) P.t[i] = pose_dev[9 + i];
    }
    int64_t n = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int v = (int)(i / W), u = (int)(i - (int64_t)v * W);
        float d = depth[i];
        const float qnan = __builtin_nanf("");
        float ox = qnan, oy = qnan, oz = qnan, nx = qnan, ny = qnan, nz = qnan;
        if (d == d) {
            float x = ((float)u - cx) / fx * d;
            float y = ((float)v - cy) / fy * d;
            float z = d;
            // ((r0*x + r1*y) + r2*z) + t, every op rounded (synthetic.transform_points states the same order)
            ox = ((P.r[0] * x + P.r[1] * y) + P.r[2] * z) + P.t[0];
            oy = ((P.r[3] * x + P.r[4] * y) + P.r[5] * z) + P.t[1];
            oz = ((P.r[6] * x + P.r[7] * y) + P.r[8] * z) + P.t[2];
            if (ncam) {
                float a = ncam[i * 3 + 0], b = ncam[i * 3 + 1], c = ncam[i * 3 + 2];
                nx = (P.r[0] * a + P.r[1] * b) + P.r[2] * c;
                ny = (P.r[3] * a + P.r[4] * b) + P.r[5] * c;
                nz = (P.r[6] * a + P.r[7] * b) + P.r[8] * c;
            }
        }
        xyz[i * 3 + 0] = ox; xyz[i * 3 + 1] = oy; xyz[i * 3 + 2] = oz;
        if (nrm) { nrm[i * 3 + 0] = nx; nrm[i * 3 + 1] = ny; nrm[i * 3 + 2] = nz; }
    }
}

// ext/imgproc/imgproc.cu:98-141
__global__ void __launch_bounds__(DIF_BLOCK) k_normal_weight(const float* __restrict__ pc, float* __restrict__ out, int H, int W) {
    int64_t n = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int v = (int)(i / W), u = (int)(i - (int64_t)v * W);
        float* o = out + i * 4;
        if (v < 1 || v > H - 2 || u < 1 || u > W - 2) { o[3] = -1.0f; continue; }
        const float* c = pc + i * 3;
        if (c[2] <= 1e-6) { o[3] = -1.0f; continue; }
        const float* xp = pc + (i + 1) * 3; const float* xm = pc + (i - 1) * 3;
        const float* yp = pc + (i + W) * 3; const float* ym = pc + (i - W) * 3;
        if (xp[2] < 1e-6 || xm[2] < 1e-6 || yp[2] < 1e-6 || ym[2] < 1e-6) { o[3] = -1.0f; continue; }
        float dxx = xp[0] - xm[0], dxy = xp[1] - xm[1], dxz = xp[2] - xm[2];
        float dyx = yp[0] - ym[0], dyy = yp[1] - ym[1], dyz = yp[2] - ym[2];
        float nx = dyy * dxz - dyz * dxy, ny = dyz * dxx - dyx * dxz, nz = dyx * dxy - dyy * dxx;   // cross(diff_y, diff_x)
        float len = sqrtf(nx * nx + ny * ny + nz * nz);
        if (len < 1e-6) { o[3] = -1.0f; continue; }
        nx /= len; ny /= len; nz /= len;
        float theta = acosf(nz);
        float td = theta / (0.5f * 3.14159f - theta);
        float wgt = (0.0012f + 0.0019f * (c[2] - 0.4f) * (c[2] - 0.4f) + 0.0001f / sqrtf(c[2]) * td * td);
        o[0] = nx; o[1] = ny; o[2] = nz; o[3] = 1.0f / wgt;
    }
}

// ---- 8f-2: image-space preprocessing next to the path -------------------------------------------------------------------
// filter_depth (ext/imgproc/imgproc.cu:48-94): 5x5 bilateral filter whose range sigma follows the depth-noise model;
// border pixels (2 px) are left untouched, depth < 1e-6 -> 0.
__global__ void __launch_bounds__(DIF_BLOCK) k_filter_depth(const float* __restrict__ in, float* __restrict__ out, int H, int W) {
    const float sig_l2 = 1.2232f * 1.2232f;                  // MEAN_SIGMA_L^2
    int64_t n = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int v = (int)(i / W), u = (int)(i - (int64_t)v * W);
        if (v < 2 || v >= H - 2 || u < 2 || u >= W - 2) continue;
        float z = in[i];
        if (z < 1e-6) { out[i] = 0.0f; continue; }
        float sigma_z = 1.0f / (0.0012f + 0.0019f * (z - 0.4f) * (z - 0.4f) + 0.0001f / sqrtf(z) * 0.25f);
        float w_sum = 0.0f, acc = 0.0f;
        for (int di = -2; di <= 2; ++di)
            for (int dj = -2; dj <= 2; ++dj) {
                float nz = in[i + (int64_t)di * W + dj];
                if (nz < 1e-6) continue;
                float dz = (nz - z) * (nz - z);
                float wgt = expf(-0.5f * ((float)(abs(di) + abs(dj)) * sig_l2 + dz * sigma_z * sigma_z));
                w_sum += wgt;
                acc += wgt * nz;
            }
        out[i] = acc / w_sum;
    }
}

// point_box_filter (system/tracker.py:13-23): mean point / mean normal per voxel_size box, boxes in ascending linear id
// (x fastest).  Bounds -> box bitmap -> ordered ranks -> order-independent fixed-point sums -> means.
struct BoxGrid { float minb[3]; int n[3]; };

__device__ __forceinline__ unsigned f2ord(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

__global__ void __launch_bounds__(DIF_BLOCK) k_pbf_bounds(const float* __restrict__ pts, int64_t N, unsigned* __restrict__ mm /* [6]: min xyz, max xyz (ordered uint) */) {
    unsigned lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x)
#pragma unroll
        for (int a = 0; a < 3; ++a) { unsigned o = f2ord(pts[i * 3 + a]); lo[a] = min(lo[a], o); hi[a] = max(hi[a], o); }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int d = 32; d >= 1; d >>= 1) { lo[a] = min(lo[a], (unsigned)__shfl_xor((int)lo[a], d)); hi[a] = max(hi[a], (unsigned)__shfl_xor((int)hi[a], d)); }
        if (lane_id() == 0) { atomicMin(mm + a, lo[a]); atomicMax(mm + 3 + a, hi[a]); }
    }
}

__device__ __forceinline__ BoxGrid pbf_grid(const unsigned* __restrict__ mm, float vs) {
    BoxGrid G;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float mn = ord2f(mm[a]) - vs * 0.5f, mx = ord2f(mm[3 + a]) + vs * 0.5f;     // tracker.py:15-16
        G.minb[a] = mn;
        G.n[a] = (int)floorf((mx - mn) / vs) + 16;                                    // tracker.py:18
    }
    return G;
}

__device__ __forceinline__ int64_t pbf_cell(const BoxGrid& G, const float* p, float vs) {
    int64_t cx = (int64_t)floorf((p[0] - G.minb[0]) / vs), cy = (int64_t)floorf((p[1] - G.minb[1]) / vs), cz = (int64_t)floorf((p[2] - G.minb[2]) / vs);
    return cx + cy * G.n[0] + cz * (int64_t)G.n[0] * G.n[1];                          // tracker.py:17,19
}

__global__ void __launch_bounds__(DIF_BLOCK) k_pbf_mark(const float* __restrict__ pts, int64_t N, float vs, const unsigned* __restrict__ mm,
                                                      uint32_t* __restrict__ bits, int64_t max_cells, int* __restrict__ status) {
    const BoxGrid G = pbf_grid(mm, vs);
    if ((int64_t)G.n[0] * G.n[1] * G.n[2] > max_cells) { if (blockIdx.x == 0 && threadIdx.x == 0) status[0] = 1; return; }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t c = pbf_cell(G, pts + i * 3, vs);
        uint32_t b = 1u << (c & 31);
        if (!(bits[c >> 5] & b)) atomicOr(bits + (c >> 5), b);
    }
}

struct BoxRankFunctor {      // exclusive prefix of popcounts per bitmap word = rank of the word's first box
    const uint32_t* bits;
    int* word_rank;
    int* out_count;
    __device__ int count(int w) const { return __popc(bits[w]); }
    __device__ void emit(int w, int offset) const { word_rank[w] = offset; }
    __device__ void finish(int total) const { out_count[0] = total; }
};

__global__ void __launch_bounds__(DIF_BLOCK) k_pbf_accumulate(const float* __restrict__ pts, const float* __restrict__ nrm, int64_t N, float vs,
                                                            const unsigned* __restrict__ mm, const uint32_t* __restrict__ bits,
                                                            const int* __restrict__ word_rank, long long* __restrict__ sums /* [boxes][8] */,
                                                            const int* __restrict__ status) {
    if (status[0]) return;
    const BoxGrid G = pbf_grid(mm, vs);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t c = pbf_cell(G, pts + i * 3, vs);
        int r = word_rank[c >> 5] + __popc(bits[c >> 5] & ((1u << (c & 31)) - 1u));
        long long* s = sums + (int64_t)r * 8;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            atomicAdd((unsigned long long*)(s + a), (unsigned long long)__float2ll_rn(pts[i * 3 + a] * 16777216.0f));       // 2^-24 fixed point
            atomicAdd((unsigned long long*)(s + 3 + a), (unsigned long long)__float2ll_rn(nrm[i * 3 + a] * 16777216.0f));
        }
        atomicAdd((unsigned long long*)(s + 6), 1ull);
    }
}

__global__ void __launch_bounds__(DIF_BLOCK) k_pbf_finish(const long long* __restrict__ sums, const int* __restrict__ n_boxes, float* __restrict__ out_pts,
                                                        float* __restrict__ out_nrm, uint32_t* __restrict__ bits, const float* __restrict__ pts, int64_t N,
                                                        float vs, const unsigned* __restrict__ mm, const int* __restrict__ status) {
    const int nb = n_boxes[0];
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < (int64_t)nb * 3; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = e / 3; int a = (int)(e - r * 3);
        const float cnt = (float)sums[r * 8 + 6];
        out_pts[e] = (float)((double)sums[r * 8 + a] * (1.0 / 16777216.0)) / cnt;
        out_nrm[e] = (float)((double)sums[r * 8 + 3 + a] * (1.0 / 16777216.0)) / cnt;
    }
    if (status[0]) return;
    const BoxGrid G = pbf_grid(mm, vs);                      // restore the bitmap to all-zero for the next call
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t c = pbf_cell(G, pts + i * 3, vs);
        bits[c >> 5] = 0u;
    }
}

// =================================================================================================================
// a9 : flat groupby_sum (ext/indexing/indexing.cu:59-109) — API parity entry; the map path uses the sorted reduction
// =================================================================================================================
__global__ void __launch_bounds__(DIF_BLOCK) k_groupby_sum(const float* __restrict__ values, const int64_t* __restrict__ idx, int64_t N,
                                                         int Lw, float* __restrict__ sum, int* __restrict__ cnt, int64_t C) {
    int64_t total = N * Lw;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t i = e / Lw;
        int l = (int)(e - i * Lw);
        int64_t g = idx[i];
        if (g < 0 || g >= C) continue;
        atomicAdd(sum + g * Lw + l, values[e]);
        if (l == 0) atomicAdd(cnt + g, 1);
    }
}

// =================================================================================================================
// a3..a6 : voxel ids, prune, allocate   (map.py:366-387)
// =================================================================================================================
// K1: per-point voxel id + per-voxel point count of this frame.  Adjacent pixels mostly fall in the same voxel, so
// equal-id runs inside a wave are aggregated with a ballot before touching memory (1 atomic per run, not per point).
__global__ void __launch_bounds__(DIF_BLOCK) k_voxel_count(Geo g, const float* __restrict__ xyz, int64_t N, int* __restrict__ pt_lin,
                                                         int* __restrict__ frame_count, int* __restrict__ counters, int px_lo, int px_hi) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // grid covers N rounded up to a wave
    int lane = lane_id();
    if (i < 4) counters[DIF_C_ALLOC_NEW + i] = 0;                   // ALLOC_NEW, M, C, ITEMS of this call
    int lin = -2;                                                    // -2: beyond N, -1: invalid point
    if (i < N) {
        float xn, yn, zn; int ix, iy, iz;
        bool ok = voxel_of(g, xyz[i * 3 + 0], xyz[i * 3 + 1], xyz[i * 3 + 2], xn, yn, zn, ix, iy, iz);
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

// K2: prune mask + candidate voxels.  mask[i] = count(voxel of i) > prune_min_vox_obs (map.py:375).  A kept point whose
// voxel has no slot marks that voxel and its 6 clamped neighbours (if empty) in the bitmap (map.py:383-386).
__global__ void __launch_bounds__(DIF_BLOCK) k_prune_mark(Geo g, int prune_min, const int* __restrict__ pt_lin, int64_t N,
                                                        const int* __restrict__ frame_count, const int64_t* __restrict__ indexer,
                                                        uint8_t* __restrict__ unq_mask, uint32_t* __restrict__ bits,
                                                        int* __restrict__ counters) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int lane = lane_id();
    int lin = (i < N) ? pt_lin[i] : -2;
    bool keep = false;
    if (lin >= 0) keep = (prune_min > 0) ? (frame_count[lin] > prune_min) : true;
    if (i < N) unq_mask[i] = keep ? 1 : 0;
    int prev = __shfl_up(lin, 1);
    bool head = (lane == 0) || (prev != lin);
    if (head && keep && indexer[lin] == -1) {
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
#pragma unroll
        for (int c = 0; c < 7; ++c) {
            int v = cand[c];
            if (indexer[v] != -1) continue;
            uint32_t b = 1u << (v & 31);
            if (!(bits[v >> 5] & b)) atomicOr(bits + (v >> 5), b);
        }
    }
}

// K3: ordered compaction of the candidate bitmap -> slots n_occupied, n_occupied+1, ... in ASCENDING lin order
// (torch.unique order, map.py:385-387, 310-319).  Clears the bitmap as it goes.
struct AllocFunctor {
    uint32_t* bits;
    int64_t* indexer;
    int64_t* pos;
    int* counters;
    int64_t capacity;
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
            }
            ++base;
        }
    }
    __device__ void finish(int total) const { counters[DIF_C_ALLOC_NEW] = total; }
};

// K4: (i) commit n_occupied += newly allocated (all pass-2 blocks of K3 have read the old value by now),
// (ii) restore frame_count to zero, (iii) focus mask + 8-offset gather keys (map.py:389-433).
// Key of pair (offset o, point i), stored at o*N + i (the reference's concatenation order): slot of the neighbour voxel
// if that voxel is in the encode set {obs_count < encoder_count_th}, else DIF_INVALID_KEY.  Rows per slot are counted here
// (seg_cnt = the reference's `pcounts`, map.py:437-439) with one atomic per distinct slot per wave.
__device__ __forceinline__ bool in_encode_set(int64_t slot, const float* __restrict__ obs, float th) { return slot >= 0 && obs[slot] < th; }

// Wave-aggregated "fetch-add 1" on counter[key] for every lane whose key is valid; returns the lane's unique offset
// (base + rank among the lanes of the wave that share the key).  One atomic per distinct key per wave.
__device__ __forceinline__ int wave_grouped_fetch_add(int* __restrict__ counter, uint32_t key, bool valid) {
    const int lane = lane_id();
    int result = 0;
    unsigned long long todo = __ballot(valid);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t k0 = (uint32_t)__shfl((int)key, leader);
        const unsigned long long same = __ballot(valid && key == k0);
        if (valid && key == k0) {
            int base = 0;
            if (lane == leader) base = atomicAdd(counter + k0, __popcll(same));
            base = __shfl(base, leader);
            result = base + __popcll(same & ((1ull << lane) - 1ull));
        }
        todo &= ~same;
    }
    return result;
}

// Same grouping, fire-and-forget: nobody waits for the atomic's return value.
__device__ __forceinline__ void wave_grouped_add(int* __restrict__ counter, uint32_t key, bool valid) {
    const int lane = lane_id();
    unsigned long long todo = __ballot(valid);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t k0 = (uint32_t)__shfl((int)key, leader);
        const unsigned long long same = __ballot(valid && key == k0);
        if (lane == leader) atomicAdd(counter + k0, __popcll(same));
        todo &= ~same;
    }
}

__global__ void __launch_bounds__(DIF_BLOCK) k_focus_gather(Geo g, float enc_th, const float* __restrict__ xyz, const int* __restrict__ pt_lin,
                                                          const uint8_t* __restrict__ unq_mask, int64_t N, int* __restrict__ frame_count,
                                                          const int64_t* __restrict__ indexer, const float* __restrict__ obs,
                                                          uint32_t* __restrict__ pair_key, int* __restrict__ seg_cnt,
                                                          int* __restrict__ counters, int64_t capacity) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // grid covers N rounded up to whole waves
    if (i == 0) {
        int n = counters[DIF_C_N_OCCUPIED] + counters[DIF_C_ALLOC_NEW];
        if (n > capacity) { n = (int)capacity; counters[DIF_C_OVERFLOW] = 1; }
        counters[DIF_C_N_OCCUPIED] = n;
    }
    const int lin = (i < N) ? pt_lin[i] : -1;
    uint32_t key[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) key[o] = DIF_INVALID_KEY;
    if (lin >= 0) {
        frame_count[lin] = 0;
        if (unq_mask[i]) {
            float xn, yn, zn; int ix, iy, iz;
            voxel_of(g, xyz[i * 3 + 0], xyz[i * 3 + 1], xyz[i * 3 + 2], xn, yn, zn, ix, iy, iz);
            // get_pruned_surface: own voxel in expand(encode set) <=> own voxel or an in-grid 6-neighbour is in the set
            bool focus = in_encode_set(indexer[lin], obs, enc_th);
            if (!focus && ix > 0) focus = in_encode_set(indexer[lin - g.ny * g.nz], obs, enc_th);
            if (!focus && ix < g.nx - 1) focus = in_encode_set(indexer[lin + g.ny * g.nz], obs, enc_th);
            if (!focus && iy > 0) focus = in_encode_set(indexer[lin - g.nz], obs, enc_th);
            if (!focus && iy < g.ny - 1) focus = in_encode_set(indexer[lin + g.nz], obs, enc_th);
            if (!focus && iz > 0) focus = in_encode_set(indexer[lin - 1], obs, enc_th);
            if (!focus && iz < g.nz - 1) focus = in_encode_set(indexer[lin + 1], obs, enc_th);
            if (focus) {
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    float ox = (o & 4) ? 0.5f : -0.5f, oy = (o & 2) ? 0.5f : -0.5f, oz = (o & 1) ? 0.5f : -0.5f;   // map.py:186-189
                    int gx = clampi((int)(ceilf(xn + ox) - 1.0f), 0, g.nx - 1);                                   // map.py:422-424
                    int gy = clampi((int)(ceilf(yn + oy) - 1.0f), 0, g.ny - 1);
                    int gz = clampi((int)(ceilf(zn + oz) - 1.0f), 0, g.nz - 1);
                    int64_t slot = indexer[linearize(g, gx, gy, gz)];
                    if (in_encode_set(slot, obs, enc_th)) key[o] = (uint32_t)slot;
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        if (i < N) pair_key[(int64_t)o * N + i] = key[o];
        wave_grouped_add(seg_cnt, key[o], key[o] != DIF_INVALID_KEY);
    }
}

// K5: per-slot encoder work items (ceil(cnt / ITEM_ROWS)), exclusive scan over slots, item -> slot table.  A slot's rows
// live in the row table at [item_start*ITEM_ROWS, ...) (padded to whole items), so one scan yields both.
struct ItemFunctor {
    const int* seg_cnt;
    int* item_start;
    int* item_slot;
    int* counters;
    int64_t max_items;
    __device__ int count(int s) const { return (seg_cnt[s] + ITEM_ROWS - 1) / ITEM_ROWS; }
    __device__ void emit(int s, int offset) const {
        int n = (seg_cnt[s] + ITEM_ROWS - 1) / ITEM_ROWS;
        item_start[s] = offset;
        if ((int64_t)offset + n > max_items) { counters[DIF_C_OVERFLOW] = 4; return; }
        for (int k = 0; k < n; ++k) item_slot[offset + k] = s;
    }
    __device__ void finish(int total) const { counters[DIF_C_ITEMS] = (total > max_items) ? (int)max_items : total; }
};

// K6: place every valid (offset, point) pair into its slot's rows.  Order inside a slot is arrival order — harmless, because
// the per-voxel sum is accumulated in exact fixed point (order-independent, see k_encode).
__global__ void __launch_bounds__(DIF_BLOCK) k_scatter_rows(const uint32_t* __restrict__ pair_key, int64_t n_pairs, const int* __restrict__ item_start,
                                                          int* __restrict__ seg_cursor, uint32_t* __restrict__ row_val, int64_t max_rows) {
    const int64_t n_pad = (n_pairs + 63) / 64 * 64;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_pad; j += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t key = (j < n_pairs) ? pair_key[j] : DIF_INVALID_KEY;
        const bool valid = key != DIF_INVALID_KEY;
        const int r = wave_grouped_fetch_add(seg_cursor, key, valid);
        if (valid) {
            const int64_t pos = (int64_t)item_start[key] * ITEM_ROWS + r;
            if (pos < max_rows) row_val[pos] = (uint32_t)j;
        }
    }
}

// =================================================================================================================
// a7..a9 : gather + encoder (MFMA) + per-voxel sums
// =================================================================================================================
// Persistent: one 512-thread workgroup per CU keeps the 107 KB of packed encoder weights in LDS; each wave pulls work
// items (slot, 32 rows), runs the tile through the MFMA chain and reduces the 29 output features over the rows.
// The reduction is done in 2^-30 FIXED POINT (int64): integer addition is associative, so the per-voxel sum does not
// depend on row order, tile grouping or the order partials are added in => bit-reproducible, and more accurate than an
// fp32 running sum (the reference sums with float atomics in arbitrary order, indexing.cu:59-71).
#define DIF_FIX_SCALE 1073741824.0f          /* 2^30: |enc| < 2^12 and < 2^21 rows per voxel keep the sum inside int64 */
__global__ void __launch_bounds__(512, 2)
k_encode(Geo g, const float* __restrict__ wblob, const float* __restrict__ xyz, const float* __restrict__ normal, int64_t N,
         const uint32_t* __restrict__ row_val, const int* __restrict__ seg_cnt, const int* __restrict__ item_start,
         const int* __restrict__ item_slot, const int* __restrict__ counters, long long* __restrict__ partial /* [items][32] */) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    stage_weights(lds, wblob, ENC_FLOATS);
    const int lane = lane_id(), half = lane >> 5, col = lane & 31;
    // work item w goes to wave (w / #blocks) of block (w % #blocks): a partly filled launch spreads over all CUs and SIMDs first
    const int wave = (int)((threadIdx.x >> 6) * gridDim.x + blockIdx.x);
    const int nwaves = (int)(gridDim.x * (blockDim.x >> 6));
    const int n_items = counters[DIF_C_ITEMS];
    for (int item = wave; item < n_items; item += nwaves) {
        const int slot = item_slot[item];
        const int chunk = item - item_start[slot];
        const bool live = chunk * ITEM_ROWS + col < seg_cnt[slot];
        float x0 = 0.f, x1 = 0.f, x2 = 0.f;
        if (live) {
            uint32_t v = row_val[(int64_t)item * ITEM_ROWS + col];
            int o = 0;
#pragma unroll
            for (int k = 1; k < 8; ++k) o += ((int64_t)v >= (int64_t)k * N) ? 1 : 0;
            int64_t i = (int64_t)v - (int64_t)o * N;
            float xn = normalize1(xyz[i * 3 + 0], g.bx, g.vs);
            float yn = normalize1(xyz[i * 3 + 1], g.by, g.vs);
            float zn = normalize1(xyz[i * 3 + 2], g.bz, g.vs);
            float ox = (o & 4) ? 0.5f : -0.5f, oy = (o & 2) ? 0.5f : -0.5f, oz = (o & 1) ? 0.5f : -0.5f;
            float gx = fminf(fmaxf(ceilf(xn + ox) - 1.0f, 0.0f), (float)(g.nx - 1));
            float gy = fminf(fmaxf(ceilf(yn + oy) - 1.0f, 0.0f), (float)(g.ny - 1));
            float gz = fminf(fmaxf(ceilf(zn + oz) - 1.0f, 0.0f), (float)(g.nz - 1));
            float rx = (xn - gx) - 0.5f, ry = (yn - gy) - 0.5f, rz = (zn - gz) - 0.5f;      // map.py:425
            float nxv = normal[i * 3 + 0], nyv = normal[i * 3 + 1], nzv = normal[i * 3 + 2];
            x0 = half ? ry : rx;
            x1 = half ? nxv : rz;
            x2 = half ? nzv : nyv;
        }
        f16v out = encoder_tile(lds, x0, x1, x2, lane);
        long long* p = partial + (int64_t)item * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            long long v = live ? __float2ll_rn(out[r] * DIF_FIX_SCALE) : 0ll;
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 16);
            if (col == 0) p[(r & 3) + 8 * (r >> 2) + 4 * half] = v;
        }
    }
}

// a10: fusion update (map.py:448-452).  One 32-lane group per slot.
__global__ void __launch_bounds__(DIF_BLOCK) k_fuse(const long long* __restrict__ partial, const int* __restrict__ item_start, int* __restrict__ seg_cnt,
                                                  int* __restrict__ seg_cursor, float* __restrict__ latent, float* __restrict__ obs,
                                                  uint8_t* __restrict__ dirty, int* __restrict__ counters) {
    __shared__ int smem[8];
    const int n_occ = counters[DIF_C_N_OCCUPIED];
    const int grp = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 5), ngrp = (int)((gridDim.x * blockDim.x) >> 5);
    const int f = threadIdx.x & 31;
    int updated = 0, rows = 0;
    for (int s = grp; s < n_occ; s += ngrp) {
        int cnt = seg_cnt[s];
        if (cnt <= 0) continue;
        int it0 = item_start[s], nit = (cnt + ITEM_ROWS - 1) / ITEM_ROWS;
        if (f < L) {
            long long Si = 0;
            for (int k = 0; k < nit; ++k) Si += partial[(int64_t)(it0 + k) * 32 + f];
            float S = (float)Si * (1.0f / DIF_FIX_SCALE);    // one rounding: exact integer sum -> nearest float
            float w_old = obs[s];
            float z_old = latent[(int64_t)s * L + f];
            S = S + z_old * w_old;                           // map.py:449
            float w_new = w_old + (float)cnt;                // map.py:450
            latent[(int64_t)s * L + f] = S / w_new;          // map.py:451
        }
        __builtin_amdgcn_wave_barrier();
        if (f == 31) {                                       // after every lane of the group has read obs[s]
            obs[s] = obs[s] + (float)cnt;
            dirty[s] = 1;                                    // map.py:452
            seg_cnt[s] = 0;
            seg_cursor[s] = 0;
            ++updated;
            rows += cnt;
        }
    }
    int tu = block_sum(updated, smem);
    int tr = block_sum(rows, smem);
    if (threadIdx.x == 0 && tu) { atomicAdd(counters + DIF_C_C, tu); atomicAdd(counters + DIF_C_M, tr); }
}

// =================================================================================================================
// a11 : extract — dirty list, confident neighbourhood, batch ids  (map.py:627-637)
// =================================================================================================================
// set the bitmap bits of the confident voxels among `lin` and its 6 allocated neighbours (map.py:628-631)
__device__ __forceinline__ void mark_confident_nbhd(const Geo& g, int lin, float ignore_th, const int64_t* __restrict__ indexer,
                                                    const float* __restrict__ obs, uint32_t* __restrict__ bits) {
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
#pragma unroll
    for (int c = 0; c < 7; ++c) {
        int v = cand[c];
        int64_t slot = indexer[v];
        if (slot < 0 || !(obs[slot] > ignore_th)) continue;
        uint32_t b = 1u << (v & 31);
        if (!(bits[v >> 5] & b)) atomicOr(bits + (v >> 5), b);
    }
}

// Spatial tiling: dirty HALO voxels (flag copied from their owner by the halo refresh) are not meshed here, but they pull their
// confident neighbourhood into the decoded batch exactly as they do in the single-map run (the blend of a corner depends on
// which neighbours are in the batch, mc_interp_kernel.cu:17-24).
__global__ void __launch_bounds__(DIF_BLOCK) k_mark_halo_dirty(Geo g, float ignore_th, uint8_t* __restrict__ dirty, const int64_t* __restrict__ pos,
                                                             const int64_t* __restrict__ indexer, const float* __restrict__ obs,
                                                             uint32_t* __restrict__ bits, const int* __restrict__ counters, int64_t own_lo,
                                                             int64_t own_hi) {
    const int n = counters[DIF_C_N_OCCUPIED];
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
        if (!dirty[s]) continue;
        const int64_t p = pos[s];
        if (p >= own_lo && p < own_hi) continue;
        dirty[s] = 0;
        mark_confident_nbhd(g, (int)p, ignore_th, indexer, obs, bits);
    }
}

struct DirtyFunctor {       // ordered compaction of dirty flags over slots -> valid_blocks (lin ids), clears flags;
    uint8_t* dirty;         // each dirty voxel also marks the confident voxels among itself and its 6 allocated neighbours
    const int64_t* pos;     // in the grid bitmap (map.py:628-631)
    int64_t* valid_blocks;
    int* counters;
    int no_cache;
    int64_t max_voxels;
    Geo g;
    float ignore_th;
    const int64_t* indexer;
    const float* obs;
    uint32_t* bits;
    int64_t own_lin_lo, own_lin_hi;     // only owned voxels are meshed (spatial tiling); the whole grid by default
    __device__ int count(int s) const {
        if (!(no_cache || dirty[s])) return 0;
        const int64_t p = pos[s];
        return (p >= own_lin_lo && p < own_lin_hi) ? 1 : 0;      // halo voxels are meshed by their owner
    }
    __device__ void emit(int s, int offset) const {
        dirty[s] = 0;
        if (offset >= max_voxels) return;
        const int lin = (int)pos[s];
        valid_blocks[offset] = lin;
        mark_confident_nbhd(g, lin, ignore_th, indexer, obs, bits);
    }
    __device__ void finish(int total) const {
        if (total > max_voxels) { total = (int)max_voxels; counters[DIF_C_OVERFLOW] = 2; }
        counters[DIF_C_K] = total;
    }
};

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

// decode mode: 0 = lattice (rows are (voxel b, sample s)), 1 = refine list, 2 = explicit rows, 3 = map point query
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
};

// GRAD: 256 threads = one wave per SIMD with the full 512-register budget (forward + reverse chain keep ~300 values live)
template <bool GRAD>
__global__ void __launch_bounds__(GRAD ? 256 : 512, GRAD ? 1 : 2) k_decode(DecodeArgs A, const float* __restrict__ wblob) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
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
        bool live;
        int64_t out_idx = 0;
        const float* lat_row = nullptr;
        float px = 0.f, py = 0.f, pz = 0.f;
        const float* row32 = nullptr;
        if (A.mode == 0) {
            int64_t b = tile / tiles_per_voxel;
            int s = (int)(tile - b * tiles_per_voxel) * 32 + col;
            live = s < res3;
            if (live) {
                int r = A.lat.res;
                px = A.lat.coord(s / (r * r)); py = A.lat.coord((s / r) % r); pz = A.lat.coord(s % r);
                lat_row = A.latent + (int64_t)A.occ_slot[b] * L;
                out_idx = b * res3 + s;
            }
        } else if (A.mode == 1) {
            int64_t row = tile * 32 + col;
            live = row < n_rows;
            if (live) {
                int e = A.list[row];
                int b = e / res3, s = e - b * res3, r = A.lat.res;
                px = A.lat.coord(s / (r * r)); py = A.lat.coord((s / r) % r); pz = A.lat.coord(s % r);
                lat_row = A.latent + (int64_t)A.occ_slot[b] * L;
                out_idx = e;
            }
        } else if (A.mode == 2) {
            int64_t row = tile * 32 + col;
            live = row < n_rows;
            if (live) { row32 = A.rows + row * 32; out_idx = row; }
        } else {
            int64_t row = tile * 32 + col;
            live = row < n_rows;
            if (live) {
                int64_t p = A.list[row];
                float xn, yn, zn; int ix, iy, iz;
                voxel_of(A.geo, A.xyz[p * 3 + 0], A.xyz[p * 3 + 1], A.xyz[p * 3 + 2], xn, yn, zn, ix, iy, iz);
                px = (xn - (float)ix) - 0.5f; py = (yn - (float)iy) - 0.5f; pz = (zn - (float)iz) - 0.5f;      // map.py:575
                lat_row = A.latent + A.indexer[linearize(A.geo, ix, iy, iz)] * L;
                out_idx = row;
            }
        }
        f16v xin;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int k = 2 * t + half;                     // natural k order of layer 0 (and of the skip input)
            float v = 0.0f;
            if (live) {
                if (row32) v = row32[k];
                else if (k < L) v = lat_row[k];
                else v = (k == L) ? px : ((k == L + 1) ? py : pz);
            }
            xin[t] = v;
        }
        float sdf, sd;
        if (GRAD) {
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
// one wave owns one voxel — the l^3 low samples go through the MLP and stay in LDS, the ATen-exact trilinear x2 upsample reads
// them from there, the cube is written once, and the |sdf| < 0.05 samples are appended to the global refine list with ONE
// atomic per voxel (wave prefix sum of popcounts).  Work per voxel is uniform (ceil(l^3/32) tiles), so the launch is balanced;
// the exact re-decode of the selected samples stays a separate, globally balanced launch (per-voxel counts range 0..R^3).
struct VoxelDecodeArgs {
    const int32_t* occ_slot;
    const float* latent;
    Lattice low;
    int R;
    float* cube_sdf;
    float* cube_std;
    int32_t* refine_list;
    int* counters;
};

#define VD_MAX_L3 64
#define VD_MAX_R2 64
#define VD_WAVE_LDS_FLOATS (2 * VD_MAX_L3)      /* low sdf + low std */

__global__ void __launch_bounds__(512, 2) k_decode_voxels(VoxelDecodeArgs A, const float* __restrict__ wblob) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    stage_weights(lds, wblob, DEC_LDS_FLOATS);
    const __amdgpu_buffer_rsrc_t wfwd = make_rsrc(wblob, DEC_FLOATS);
    const int lane = lane_id(), half = lane >> 5, col = lane & 31, wid = threadIdx.x >> 6;
    float* w_low_sdf = lds + ((DEC_LDS_FLOATS + 3) & ~3) + wid * VD_WAVE_LDS_FLOATS;
    float* w_low_std = w_low_sdf + VD_MAX_L3;
    const int l = A.low.res, R = A.R, l3 = l * l * l, R2 = R * R, R3 = R2 * R;
    const float scale = (float)(l - 1) / (float)(R - 1);
    const int B = A.counters[DIF_C_B];
    const int wave = (int)(wid * gridDim.x + blockIdx.x), nwaves = (int)(gridDim.x * (blockDim.x >> 6));   // spread over CUs first
    for (int b = wave; b < B; b += nwaves) {
        const float* lat_row = A.latent + (int64_t)A.occ_slot[b] * L;
        f16v xlat;                                          // latent part of the B operand: the same for every sample of the voxel
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int k = 2 * t + half;
            xlat[t] = (k < L) ? lat_row[k] : 0.0f;
        }
        // ---- low lattice -> LDS (map.py:644-653) ----
        for (int t0 = 0; t0 < l3; t0 += 32) {
            const int s = t0 + col;
            const bool live = s < l3;
            const float px = A.low.coord(s / (l * l)), py = A.low.coord((s / l) % l), pz = A.low.coord(s % l);
            f16v xin = xlat;
            if (half) { xin[14] = px; xin[15] = pz; } else { xin[15] = py; }        // k = 29 (x), 30 (y), 31 (z)
            float sdf, sd;
            decoder_tile(lds, wfwd, xin, lane, sdf, sd);
            if (live) {
                if (half == 0) w_low_sdf[s] = sdf;
                else w_low_std[s] = sd;
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        // ---- trilinear x2 + threshold (map.py:655-667): lane = (jx, jy) row of R samples along z ----
        const int64_t e0 = (int64_t)b * R3 + (int64_t)lane * R;
        unsigned sel = 0;
        if (lane < R2) {
            const int jx = lane / R, jy = lane % R;
            int x0, x1, y0, y1; float wx0, wx1, wy0, wy1;
            tri_axis(jx, l, scale, x0, x1, wx0, wx1);
            tri_axis(jy, l, scale, y0, y1, wy0, wy1);
            for (int jz = 0; jz < R; ++jz) {
                int z0, z1; float wz0, wz1;
                tri_axis(jz, l, scale, z0, z1, wz0, wz1);
                float sv = tri_sample(w_low_sdf, l, x0, x1, y0, y1, z0, z1, wx0, wx1, wy0, wy1, wz0, wz1);
                float dv = tri_sample(w_low_std, l, x0, x1, y0, y1, z0, z1, wx0, wx1, wy0, wy1, wz0, wz1);
                A.cube_sdf[e0 + jz] = -sv;
                A.cube_std[e0 + jz] = dv;
                if (fabsf(sv) < 0.05f) sel |= 1u << jz;
            }
        }
        const int c = __popc(sel);
        const int incl = wave_incl_scan(c);
        const int total = __shfl(incl, 63);
        if (total > 0) {
            int base = 0;
            if (lane == 0) base = atomicAdd(A.counters + DIF_C_VH, total);
            base = __shfl(base, 0);
            int o = base + incl - c;
            while (sel) {
                const int jz = __ffs((int)sel) - 1;
                sel &= sel - 1;
                A.refine_list[o++] = (int32_t)(e0 + jz);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// =================================================================================================================
// a15 : sparse marching cubes with cross-voxel std-weighted blending (ext/marching_cubes/mc_interp_kernel.cu:7-320)
// =================================================================================================================
struct McArgs {
    const int64_t* indexer; int nx, ny, nz;
    const int64_t* valid_blocks; const int* K_ptr; int64_t K_static;
    const int32_t* vbm; int64_t V;
    const float* cube_sdf; const float* cube_std; int R;
    float max_std;
    int64_t max_triangles;
    float* triangles; int64_t* tri_id; float* tri_std; uint8_t* tri_alive;
    int32_t* tri_count; const int32_t* tri_offset;
    const int* base_ptr;            // device: first output triangle index (mesh-cache append), or NULL
    int64_t new_limit;              // triangles this call may emit (max_n_triangles)
    int scale; float vs, bx, by, bz;
};

// batch index of voxel (bx,by,bz) or -1   (query_sdf_raw :13-24)
__device__ __forceinline__ int mc_batch_of(const McArgs& a, int bx, int by, int bz) {
    if ((unsigned)bx >= (unsigned)a.nx || (unsigned)by >= (unsigned)a.ny || (unsigned)bz >= (unsigned)a.nz) return -1;
    int64_t vec = a.indexer[((int64_t)bx * a.ny + by) * a.nz + bz];
    if (vec == -1 || vec >= a.V) return -1;
    return a.vbm[vec];
}

// get_sdf (:34-185), STD_W_SDF branch: blend of the <=8 voxels whose cubes overlap corner `c` of voxel at nb[13].
// nb: batch ids of the 3x3x3 neighbourhood (index (dx+1)*9 + (dy+1)*3 + (dz+1)).  Returns false => NaN (cell dropped).
__device__ __forceinline__ bool mc_corner(const McArgs& a, const int* nb, int r, int cx, int cy, int cz, float& sdf, float& sd) {
    const int R = a.R;
    const int rbound = (r - 1) / 2, rstart = r / 2;
    const float rmid = (float)r / 2.0f;
    int c[3] = {cx, cy, cz};
    int dm[3], dp[3], im[3], ip[3], zero[3];
    float wm[3], wp[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        if (c[ax] <= rbound) {
            dm[ax] = -1; im[ax] = c[ax] + rstart + r; dp[ax] = 0; ip[ax] = c[ax] + rstart;
            wp[ax] = (float)c[ax] + rmid; wm[ax] = rmid - (float)c[ax];
            zero[ax] = 1;
        } else {
            dm[ax] = 0; im[ax] = c[ax] + rstart; dp[ax] = 1; ip[ax] = c[ax] + rstart - r;
            wp[ax] = (float)c[ax] - rmid; wm[ax] = rmid + (float)r - (float)c[ax];
            zero[ax] = 0;
        }
        wm[ax] /= (float)r; wp[ax] /= (float)r;
    }
    const int zero_det = zero[0] * 4 + zero[1] * 2 + zero[2];
    float ts = 0.0f, tw = 0.0f, tsd = 0.0f, twd = 0.0f;     // total_sdf.x, total_weight.x, total_sdf.y, total_weight.y
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int sx = (k >> 2) & 1, sy = (k >> 1) & 1, sz = k & 1;
        const int ddx = sx ? dp[0] : dm[0], ddy = sy ? dp[1] : dm[1], ddz = sz ? dp[2] : dm[2];
        const int b = nb[(ddx + 1) * 9 + (ddy + 1) * 3 + (ddz + 1)];
        float s = __builtin_nanf(""), d = 0.0f;
        if (b >= 0) {
            const int64_t off = (((int64_t)b * R + (sx ? ip[0] : im[0])) * R + (sy ? ip[1] : im[1])) * R + (sz ? ip[2] : im[2]);
            s = a.cube_sdf[off];
            d = a.cube_std[off];
        }
        const float w = (sx ? wp[0] : wm[0]) * (sy ? wp[1] : wm[1]) * (sz ? wp[2] : wm[2]);
        if (s == s) {
            ts += s * w * d; tw += w * d;
            tsd += w * d;    twd += w;
        } else if (zero_det == k) {
            return false;
        }
    }
    sdf = ts / tw;
    sd = tsd / twd;
    return sdf == sdf;
}

struct V4 { float x, y, z, w; };

__device__ __forceinline__ V4 mc_interp(const float* p1, const float* p2, float s1, float s2, float v1, float v2) {   // sdf_interp :187-200
    if (fabsf(0.0f - v1) < 1.0e-5f) return V4{p1[0], p1[1], p1[2], s1};
    if (fabsf(0.0f - v2) < 1.0e-5f) return V4{p2[0], p2[1], p2[2], s2};
    if (fabsf(v1 - v2) < 1.0e-5f) return V4{p1[0], p1[1], p1[2], s1};
    float w2 = (0.0f - v1) / (v2 - v1);
    float w1 = 1 - w2;
    return V4{p1[0] * w1 + p2[0] * w2, p1[1] * w1 + p2[1] * w2, p1[2] * w1 + p2[2] * w2, s1 * w1 + s2 * w2};
}

// One wave per dirty voxel.  Phase 1: the (r+1)^3 blended corner values are computed ONCE into LDS (the reference
// recomputes each corner for up to 8 cells).  Phase 2: lane = cell; EMIT=false counts the triangles that survive
// max_std, EMIT=true writes them at tri_offset[k] + wave-prefix (canonical order: voxel, cell, table order).
template <bool EMIT>
__global__ void __launch_bounds__(DIF_BLOCK) k_marching_cubes(McArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int r = a.R / 2, r1 = r + 1, nc = r1 * r1 * r1, r3 = r * r * r;
    const int lane = lane_id(), wid = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    float* c_sdf = lds + (size_t)wid * (2 * nc + 32);
    float* c_std = c_sdf + nc;
    int* nb = reinterpret_cast<int*>(c_std + nc);            // 27 (+pad)
    const int64_t K = a.K_ptr ? (int64_t)(*a.K_ptr) : a.K_static;
    const float sbs = 1.0f / (float)r;
    for (int64_t k = (int64_t)blockIdx.x * wpb + wid; k < K; k += (int64_t)gridDim.x * wpb) {
        const int64_t vb = a.valid_blocks[k];
        const int bx = (int)((vb / ((int64_t)a.ny * a.nz)) % a.nx), by = (int)((vb / a.nz) % a.ny), bz = (int)(vb % a.nz);
        if (lane < 27) nb[lane] = mc_batch_of(a, bx + lane / 9 - 1, by + (lane / 3) % 3 - 1, bz + lane % 3 - 1);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0): nb[] visible to the whole wave
        for (int c = lane; c < nc; c += 64) {
            float s, d;
            bool ok = mc_corner(a, nb, r, c / (r1 * r1), (c / r1) % r1, c % r1, s, d);
            c_sdf[c] = ok ? s : __builtin_nanf("");
            c_std[c] = ok ? d : 0.0f;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        int voxel_total = 0;
        for (int s0 = 0; s0 < r3; s0 += 64) {
            const int s = s0 + lane;
            int ntri = 0;
            V4 vl[12];
            int cube_type = 0;
            if (s < r3) {
                const int rx = s / (r * r), ry = (s / r) % r, rz = s % r;
                float val[8], sdv[8], pts[8][3];
                bool dropped = false;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int dx = (q == 1 || q == 2 || q == 5 || q == 6), dy = (q == 2 || q == 3 || q == 6 || q == 7), dz = (q >= 4);
                    const int ci = ((rx + dx) * r1 + (ry + dy)) * r1 + (rz + dz);
                    val[q] = c_sdf[ci]; sdv[q] = c_std[ci];
                    dropped |= !(val[q] == val[q]);
                    pts[q][0] = (float)bx + (float)(rx + dx) * sbs;
                    pts[q][1] = (float)by + (float)(ry + dy) * sbs;
                    pts[q][2] = (float)bz + (float)(rz + dz) * sbs;
                }
                if (!dropped) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) cube_type |= (val[q] < 0.0f) ? (1 << q) : 0;
                    const int edge_config = c_mc_edge_table[cube_type];
                    if (edge_config) {
                        const int ea[12] = {0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3}, eb[12] = {1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7};
#pragma unroll
                        for (int e = 0; e < 12; ++e)
                            if (edge_config & (1 << e)) vl[e] = mc_interp(pts[ea[e]], pts[eb[e]], sdv[ea[e]], sdv[eb[e]], val[ea[e]], val[eb[e]]);
                        for (int i = 0; c_mc_tri_table[cube_type][i] != -1; i += 3) {
                            float w0 = vl[c_mc_tri_table[cube_type][i]].w, w1 = vl[c_mc_tri_table[cube_type][i + 1]].w,
                                  w2 = vl[c_mc_tri_table[cube_type][i + 2]].w;
                            if (w0 > a.max_std || w1 > a.max_std || w2 > a.max_std) continue;     // :304
                            ++ntri;
                        }
                    } else {
                        cube_type = 0;
                    }
                } else {
                    cube_type = 0;
                }
            }
            const int incl = wave_incl_scan(ntri);
            const int chunk_total = __shfl(incl, 63);
            if (EMIT && ntri > 0) {
                int64_t tl = (int64_t)a.tri_offset[k] + voxel_total + (incl - ntri);      // index among this call's triangles
                int64_t t = tl + (a.base_ptr ? (int64_t)(*a.base_ptr) : 0);
                for (int i = 0; c_mc_tri_table[cube_type][i] != -1; i += 3) {
                    V4 v0 = vl[c_mc_tri_table[cube_type][i]], v1 = vl[c_mc_tri_table[cube_type][i + 1]], v2 = vl[c_mc_tri_table[cube_type][i + 2]];
                    if (v0.w > a.max_std || v1.w > a.max_std || v2.w > a.max_std) continue;
                    if (tl < a.new_limit && t < a.max_triangles) {
                        V4 vv[3] = {v0, v1, v2};
#pragma unroll
                        for (int vi = 0; vi < 3; ++vi) {
                            float x = vv[vi].x, y = vv[vi].y, z = vv[vi].z;
                            if (a.scale) { x = x * a.vs + a.bx; y = y * a.vs + a.by; z = z * a.vs + a.bz; }   // map.py:698
                            a.triangles[(t * 3 + vi) * 3 + 0] = x;
                            a.triangles[(t * 3 + vi) * 3 + 1] = y;
                            a.triangles[(t * 3 + vi) * 3 + 2] = z;
                            a.tri_std[t * 3 + vi] = vv[vi].w;
                        }
                        a.tri_id[t] = vb;
                        if (a.tri_alive) a.tri_alive[t] = 1;
                    }
                    ++t; ++tl;
                }
            }
            voxel_total += chunk_total;
        }
        if (!EMIT && lane == 0) a.tri_count[k] = voxel_total;
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- a16 : device-resident mesh cache as an append-only log (map.py:703-714) -----------------------------------------
// A voxel that produced >= 1 new triangle replaces its previous batch (the reference drops cached triangles whose voxel id
// occurs among the new ones, map.py:708-709): mark the old batch dead, point the voxel at its new batch.
__global__ void __launch_bounds__(DIF_BLOCK) k_log_replace(const int64_t* __restrict__ valid_blocks, const int32_t* __restrict__ tri_count,
                                                         const int32_t* __restrict__ tri_offset, const int64_t* __restrict__ indexer,
                                                         int32_t* __restrict__ tri_start, int32_t* __restrict__ tri_n, uint8_t* __restrict__ alive,
                                                         int* __restrict__ counters, int64_t new_limit, int64_t capacity) {
    __shared__ int smem[8];
    const int K = counters[DIF_C_K];
    const int64_t log_n = counters[DIF_C_CACHE_T];
    int dead = 0;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < K; k += gridDim.x * blockDim.x) {
        const int c = tri_count[k];
        if (c <= 0) continue;
        const int64_t slot = indexer[valid_blocks[k]];
        const int old_n = tri_n[slot], old_s = tri_start[slot];
        for (int j = 0; j < old_n; ++j) alive[old_s + j] = 0;
        dead += old_n;
        int64_t off = tri_offset[k];
        int64_t n_new = c;
        if (off + n_new > new_limit) n_new = new_limit > off ? new_limit - off : 0;          // truncated by max_n_triangles
        if (log_n + off + n_new > capacity) n_new = capacity > log_n + off ? capacity - (log_n + off) : 0;
        tri_start[slot] = (int)(log_n + off);
        tri_n[slot] = (int)n_new;
    }
    dead = block_sum(dead, smem);
    if (threadIdx.x == 0 && dead) atomicAdd(counters + DIF_C_CACHE_DEAD, dead);
}

struct CacheLiveFunctor {       // ordered compaction of the live log entries
    const float* src_tri; const int64_t* src_id; const float* src_std; const uint8_t* alive;
    float* dst_tri; int64_t* dst_id; float* dst_std;
    int64_t out_capacity;
    int* counters;
    __device__ int count(int t) const { return alive[t] ? 1 : 0; }
    __device__ void emit(int t, int offset) const {
        if (offset >= out_capacity) return;
#pragma unroll
        for (int i = 0; i < 9; ++i) dst_tri[(int64_t)offset * 9 + i] = src_tri[(int64_t)t * 9 + i];
#pragma unroll
        for (int i = 0; i < 3; ++i) dst_std[(int64_t)offset * 3 + i] = src_std[(int64_t)t * 3 + i];
        dst_id[offset] = src_id[t];
    }
    __device__ void finish(int total) const { counters[DIF_C_CACHE_LIVE] = total > out_capacity ? (int)out_capacity : total; }
};

__global__ void __launch_bounds__(DIF_BLOCK) k_cache_reindex(const int64_t* __restrict__ id, int64_t n, const int64_t* __restrict__ indexer,
                                                           int32_t* __restrict__ tri_start, int32_t* __restrict__ tri_n, uint8_t* __restrict__ alive,
                                                           int* __restrict__ counters) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        alive[t] = 1;
        const int64_t v = id[t];
        const int64_t slot = indexer[v];
        if (slot < 0) continue;
        if (t == 0 || id[t - 1] != v) tri_start[slot] = (int)t;            // a live voxel owns exactly one contiguous batch
        atomicAdd(tri_n + slot, 1);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        counters[DIF_C_CACHE_T] = (int)n;
        counters[DIF_C_CACHE_KEPT] = (int)n;
        counters[DIF_C_CACHE_DEAD] = 0;
    }
}

// end of extract: clear the batch map, publish the log length
__global__ void __launch_bounds__(DIF_BLOCK) k_extract_finish(const int32_t* __restrict__ occ_slot, int32_t* __restrict__ vbm,
                                                            int* __restrict__ counters, int64_t new_limit, int64_t capacity) {
    const int B = counters[DIF_C_B];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) vbm[occ_slot[i]] = -1;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int64_t n_new = counters[DIF_C_T];
        if (n_new > new_limit) n_new = new_limit;
        const int64_t old_n = counters[DIF_C_CACHE_T];
        int64_t tot = old_n + n_new;
        if (tot > capacity) { tot = capacity; counters[DIF_C_OVERFLOW] = 5; }
        counters[DIF_C_CACHE_KEPT] = (int)old_n;
        counters[DIF_C_CACHE_T] = (int)tot;
    }
}

struct TriScanFunctor {
    const int32_t* tri_count;
    int32_t* tri_offset;
    int* counters;
    __device__ int count(int k) const { return tri_count[k]; }
    __device__ void emit(int k, int offset) const { tri_offset[k] = offset; }
    __device__ void finish(int total) const { counters[DIF_C_T] = total; }
};

// =================================================================================================================
// a17 : get_sdf — validity mask + ordered compaction of valid points  (map.py:565-573)
// =================================================================================================================
struct QueryFunctor {
    Geo g;
    float ignore_th;
    const float* xyz;
    const int64_t* indexer;
    const float* obs;
    uint8_t* mask;
    int32_t* sel;
    int* counters;
    __device__ int count(int i) const {
        float xn, yn, zn; int ix, iy, iz;
        bool ok = voxel_of(g, xyz[(int64_t)i * 3 + 0], xyz[(int64_t)i * 3 + 1], xyz[(int64_t)i * 3 + 2], xn, yn, zn, ix, iy, iz);
        if (ok) {
            int64_t slot = indexer[linearize(g, ix, iy, iz)];
            ok = slot >= 0 && obs[slot] > ignore_th;
        }
        mask[i] = ok ? 1 : 0;
        return ok ? 1 : 0;
    }
    __device__ void emit(int i, int offset) const { sel[offset] = i; }
    __device__ void finish(int total) const { counters[DIF_C_QUERY_M] = total; }
};

// =================================================================================================================
// multi-GPU merge helpers
// =================================================================================================================
struct ExportFunctor {       // ordered compaction over slots: allocated voxels with x index in [x_lo, x_hi)
    const int64_t* pos; const float* obs; const float* latent; const uint8_t* dirty;
    int32_t* rec; int64_t max_records;
    int64_t lin_lo, lin_hi;
    int raw;
    int* counters;
    __device__ int count(int s) const { int64_t p = pos[s]; return (p >= lin_lo && p < lin_hi) ? 1 : 0; }
    __device__ void emit(int s, int offset) const {
        if (offset >= max_records) return;
        int32_t* r = rec + (int64_t)offset * 32;
        const int64_t p = pos[s];
        const float w = obs[s];
        r[0] = (int32_t)p;                       // grid < 2^31 (checked by every entry point)
        r[1] = dirty[s] ? 1 : 0;                 // flags: bit 0 = awaiting re-meshing
        r[2] = __float_as_int(w);
        for (int f = 0; f < L; ++f) {
            float z = latent[(int64_t)s * L + f];
            r[3 + f] = __float_as_int(raw ? z : z * w);
        }
    }
    __device__ void finish(int total) const {
        if (total > max_records) { total = (int)max_records; counters[DIF_C_OVERFLOW] = 6; }
        counters[DIF_C_EXPORT_N] = total;
    }
};

__global__ void __launch_bounds__(DIF_BLOCK) k_merge_mark(const int32_t* __restrict__ rec, int64_t n, const int64_t* __restrict__ indexer,
                                                        uint32_t* __restrict__ bits, int64_t grid) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t lin = rec[i * 32];
        if (lin < 0 || lin >= grid) continue;
        if (indexer[lin] == -1) atomicOr(bits + (lin >> 5), 1u << (lin & 31));
    }
}

// records of one call carry distinct lin ids => plain read-modify-write, deterministic
__global__ void __launch_bounds__(DIF_BLOCK) k_merge_apply(const int32_t* __restrict__ rec, int64_t n, const int64_t* __restrict__ indexer,
                                                         float* __restrict__ latent, float* __restrict__ obs, uint8_t* __restrict__ dirty,
                                                         int* __restrict__ counters, int64_t grid, int64_t capacity, int assign) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int no = counters[DIF_C_N_OCCUPIED] + counters[DIF_C_ALLOC_NEW];
        if (no > capacity) { no = (int)capacity; counters[DIF_C_OVERFLOW] = 1; }
        counters[DIF_C_N_OCCUPIED] = no;
        counters[DIF_C_ALLOC_NEW] = 0;
    }
    const int64_t total = n * 32;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t i = e >> 5;
        int f = (int)(e & 31);
        int64_t lin = rec[i * 32];
        if (lin < 0 || lin >= grid) continue;
        int64_t s = indexer[lin];
        if (s < 0) continue;
        float w_r = __int_as_float(rec[i * 32 + 2]);
        float w_old = obs[s];
        float w_new = assign ? w_r : w_old + w_r;
        if (f < L) {
            float pay = __int_as_float(rec[i * 32 + 3 + f]);
            float z = latent[s * L + f];
            if (assign) latent[s * L + f] = pay;
            else if (w_new > 0.0f) latent[s * L + f] = (z * w_old + pay) / w_new;
        }
        __builtin_amdgcn_wave_barrier();
        if (f == 31) {
            obs[s] = w_new;
            if (assign) dirty[s] = (uint8_t)(rec[i * 32 + 1] & 1);
            else if (w_r > 0.0f) dirty[s] = 1;
        }
    }
}

}  // namespace

// =================================================================================================================
// C ABI
// =================================================================================================================
extern "C" {

int dif_version(void) { return DIF_VERSION; }

int dif_unproject(const float* depth, float* pc, int32_t H, int32_t W, float fx, float fy, float cx, float cy, void* stream) {
    if (!depth || !pc || H <= 0 || W <= 0) return DIF_EINVAL;
    hipLaunchKernelGGL(k_unproject, dim3(grid_for((int64_t)H * W)), dim3(DIF_BLOCK), 0, (hipStream_t)stream, depth, pc, H, W, fx, fy, cx, cy);
    DIF_CHECK_LAUNCH();
    return DIF_OK;
}

int dif_unproject_transform(const float* depth, const float* normal_cam, float* xyz_world, float* normal_world, int32_t H, int32_t W,
                            float fx, float fy, float cx, float cy, const float* R, const float* t, void* stream) {
    if (!depth || !xyz_world || !R || !t || H <= 0 || W <= 0) return DIF_EINVAL;
    if ((normal_cam == nullptr) != (normal_world == nullptr)) return DIF_EINVAL;
    Pose P;
    for (int i = 0; i < 9; ++i) P.r[i] = R[i];
    for (int i = 0; i < 3; ++i) P.t[i] = t[i];
    hipLaunchKernelGGL(k_unproject_transform, dim3(grid_for((int64_t)H * W)), dim3(DIF_BLOCK), 0, (hipStream_t)stream, depth, normal_cam,
                       xyz_world, normal_world, H, W, fx, fy, cx, cy, P, (const float*)nullptr);
    DIF_CHECK_LAUNCH();
    return DIF_OK;
}

int dif_unproject_transform_dev(const float* depth, const float* normal_cam, float* xyz_world, float* normal_world, int32_t H, int32_t W,
                                float fx, float fy, float cx, float cy, const float* pose_dev, void* stream) {
    if (!depth || !xyz_world || !pose_dev || H <= 0 || W <= 0) return DIF_EINVAL;
    if ((normal_cam == nullptr) != (normal_world == nullptr)) return DIF_EINVAL;
    Pose P = {};
    hipLaunchKernelGGL(k_unproject_transform, dim3(grid_for((int64_t)H * W)), dim3(DIF_BLOCK), 0, (hipStream_t)stream, depth, normal_cam,
                       xyz_world, normal_world, H, W, fx, fy, cx, cy, P, pose_dev);
    DIF_CHECK_LAUNCH();
    return DIF_OK;
}

int dif_compute_normal_weight(const float* pc, float* normal_weight, int32_t H, int32_t W, void* stream) {
    if (!pc || !normal_weight || H <= 0 || W <= 0) return DIF_EINVAL;
    hipLaunchKernelGGL(k_normal_weight, dim3(grid_for((int64_t)H * W)), dim3(DIF_BLOCK), 0, (hipStream_t)stream, pc, normal_weight, H, W);
    DIF_CHECK_LAUNCH();
    return DIF_OK;
}

int dif_filter_depth(const float* depth_in, float* depth_out, int32_t H, int32_t W, void* stream) {
    if (!depth_in || !depth_out || H <= 0 || W <= 0) return DIF_EINVAL;
    hipLaunchKernelGGL(k_filter_depth, dim3(grid_for((int64_t)H * W)), dim3(DIF_BLOCK), 0, (hipStream_t)stream, depth_in, depth_out, H, W);
    DIF_CHECK_LAUNCH();
    return DIF_OK;
}

int dif_point_box_filter(const float* points, const float* normals, int64_t N, float voxel_size, float* out_points, float* out_normals,
                         int32_t* out_count, uint32_t* bits, int64_t max_cells, int32_t* word_rank, int64_t* sums, int32_t* scratch, void* stream) {
    if (N < 0 || !(voxel_size > 0.0f) || max_cells <= 0 || max_cells >= ((int64_t)1 << 36)) return DIF_EINVAL;
    if (!out_count || !scratch) return DIF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) return hipMemsetAsync(out_count, 0, sizeof(int), s) == hipSuccess ? DIF_OK : DIF_ELAUNCH;
    if (!points || !normals || !out_points || !out_normals || !bits || !word_rank || !sums || N >= ((int64_t)1 << 31)) return DIF_EINVAL;
    // scratch: [0..4095] scan block totals, [4096..4101] ordered-uint bounds, [4102] status
    unsigned* mm = (unsigned*)(scratch + 4096);
    int* status = scratch + 4102;
    static const unsigned init_mm[7] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u};
    if (hipMemcpyAsync(mm, init_mm, sizeof(init_mm), hipMemcpyHostToDevice, s) != hipSuccess) return DIF_ELAUNCH;
    if (hipMemsetAsync(sums, 0, (size_t)N * 8 * sizeof(int64_t), s) != hipSuccess) return DIF_ELAUNCH;      // <= N boxes
    hipLaunchKernelGGL(k_pbf_bounds, dim3(grid_for(N, DIF_BLOCK, 1024)), dim3(DIF_BLOCK), 0, s, points, N, mm);
    hipLaunchKernelGGL(k_pbf_mark, dim3(grid_for(N)), dim3(DIF_BLOCK), 0, s, points, N, voxel_size, (const unsigned*)mm, bits, max_cells, status);
    DIF_CHECK_LAUNCH();
    BoxRankFunctor f{bits, word_rank, out_count};
    const int64_t nwords = (max_cells + 31) / 32;
    if (nwords >= ((int64_t)1 << 31)) return DIF_EINVAL;
    if (launch_scan(f, nullptr, (int)nwords, nwords, scratch, s) != DIF_OK) return DIF_ELAUNCH;
    hipLaunchKernelGGL(k_pbf_accumulate, dim3(grid_for(N)), dim3(DIF_BLOCK), 0, s, points, normals, N, voxel_size, (const unsigned*)mm,
                       (const uint32_t*)bits, (const int*)word_rank, (long long*)sums, (const int*)status);
    hipLaunchKernelGGL(k_pbf_finish, dim3(grid_for(N)), dim3(DIF_BLOCK), 0, s, (const long long*)sums, (const int*)out_count, out_points, out_normals,
                       bits, points, N, voxel_size, (const unsigned*)mm, (const int*)status);
    DIF_CHECK_LAUNCH();
    return DIF_OK;
}

int dif_groupby_sum(const float* values, const int64_t* indices, int64_t N, int32_t Lw, float* sum, int32_t* count, int64_t C, void* stream) {
    if (N < 0 || Lw <= 0 || C < 0 || (N > 0 && (!values || !indices || !sum || !count))) return DIF_EINVAL;
    if (N == 0) return DIF_OK;
    hipLaunchKernelGGL(k_groupby_sum, dim3(grid_for(N * Lw)), dim3(DIF_BLOCK), 0, (hipStream_t)stream, values, indices, N, (int)Lw, sum, count, C);
    DIF_CHECK_LAUNCH();
    return DIF_OK;
}

// ---- integrate ------------------------------------------------------------------------------------------------
// workspace carve (all offsets 256-byte aligned)
struct IntegrateWs {
    int* pt_lin;            // [N]
    uint32_t* pair_key;     // [8N]
    uint32_t* row_val;      // [max_items * ITEM_ROWS]
    int* item_slot;         // [max_items]
    long long* partial;     // [max_items][32]
    int* block_tmp;         // [4096]
    int64_t max_items;
    int64_t total_bytes;
};

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static int64_t max_items_for(int64_t N) {
    // items = sum_s ceil(cnt_s / ITEM_ROWS) <= M/ITEM_ROWS + C with M <= 8N rows.  With pruning on, every kept point shares its
    // voxel with > prune_min_vox_obs others, so the encoded voxels (26-neighbourhoods of those) number well under 2N; anything
    // beyond the bound is dropped by ItemFunctor with DIF_C_OVERFLOW = 4.
    return 8 * N / ITEM_ROWS + 2 * N + 64;
}

static int carve_integrate(int64_t N, void* base, IntegrateWs& ws) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes); return o; };
    ws.max_items = max_items_for(N);
    size_t o_lin = take((size_t)N * 4), o_key = take((size_t)8 * N * 4), o_row = take((size_t)ws.max_items * ITEM_ROWS * 4);
    size_t o_islot = take((size_t)ws.max_items * 4), o_part = take((size_t)ws.max_items * 32 * 8), o_tmp = take(4096 * 4);
    ws.total_bytes = (int64_t)off;
    if (base) {
        char* b = (char*)base;
        ws.pt_lin = (int*)(b + o_lin); ws.pair_key = (uint32_t*)(b + o_key); ws.row_val = (uint32_t*)(b + o_row);
        ws.item_slot = (int*)(b + o_islot); ws.partial = (long long*)(b + o_part); ws.block_tmp = (int*)(b + o_tmp);
    }
    return DIF_OK;
}

int64_t dif_integrate_workspace_bytes(int64_t N) {
    if (N <= 0) N = 1;
    IntegrateWs ws;
    if (carve_integrate(N, nullptr, ws) != DIF_OK) return -1;
    return ws.total_bytes;
}

int dif_integrate(const dif_map_t* map, const dif_weights_t* w, const float* xyz, const float* normal, int64_t N, uint8_t* unq_mask,
                  void* wsp, int64_t ws_bytes, void* stream_) {
    if (!map || !w || !w->enc_packed || w->enc_packed_floats != ENC_FLOATS || N < 0) return DIF_EINVAL;
    if (N == 0) return DIF_OK;
    if (!xyz || !normal || !unq_mask || !wsp) return DIF_EINVAL;
    if (8 * N >= (int64_t)1 << 31 || map->capacity >= (int64_t)DIF_INVALID_KEY) return DIF_EINVAL;
    const int64_t grid = (int64_t)map->nx * map->ny * map->nz;
    if (grid >= ((int64_t)1 << 31)) return DIF_EINVAL;
    hipStream_t s = (hipStream_t)stream_;
    IntegrateWs ws;
    if (carve_integrate(N, wsp, ws) != DIF_OK) return DIF_ELAUNCH;
    if (ws.total_bytes > ws_bytes) return DIF_ENOSPACE;
    Geo g = geo_of(map);
    int* C = map->counters;
    const int nb_pts = (int)((N + DIF_BLOCK - 1) / DIF_BLOCK);

    // k_voxel_count also zeroes the per-call counters (ALLOC_NEW, M, C, ITEMS): every kernel that writes them runs later
    const int own_lo = map->own_x_hi > map->own_x_lo ? map->own_x_lo : 0, own_hi = map->own_x_hi > map->own_x_lo ? map->own_x_hi : map->nx;
    hipLaunchKernelGGL(k_voxel_count, dim3(nb_pts), dim3(DIF_BLOCK), 0, s, g, xyz, N, ws.pt_lin, map->frame_count, C, own_lo - map->halo,
                       own_hi + map->halo);
    hipLaunchKernelGGL(k_prune_mark, dim3(nb_pts), dim3(DIF_BLOCK), 0, s, g, (int)map->prune_min_vox_obs, (const int*)ws.pt_lin, N,
                       (const int*)map->frame_count, (const int64_t*)map->indexer, unq_mask, map->grid_bits, C);
    DIF_CHECK_LAUNCH();
    {
        AllocFunctor f{map->grid_bits, map->indexer, map->latent_vecs_pos, C, map->capacity};
        int nwords = (int)((grid + 31) / 32);
        if (launch_scan(f, nullptr, nwords, nwords, ws.block_tmp, s) != DIF_OK) return DIF_ELAUNCH;
    }
    hipLaunchKernelGGL(k_focus_gather, dim3(nb_pts), dim3(DIF_BLOCK), 0, s, g, map->encoder_count_th, xyz, (const int*)ws.pt_lin,
                       (const uint8_t*)unq_mask, N, map->frame_count, (const int64_t*)map->indexer, (const float*)map->voxel_obs_count,
                       ws.pair_key, map->seg_cnt, C, map->capacity);
    DIF_CHECK_LAUNCH();
    {
        ItemFunctor f{map->seg_cnt, map->item_start, ws.item_slot, C, ws.max_items};
        if (launch_scan(f, C + DIF_C_N_OCCUPIED, 0, map->capacity, ws.block_tmp, s) != DIF_OK) return DIF_ELAUNCH;
    }
    {
        ProfScope prof(DIF_PROF_SORT, s);
        hipLaunchKernelGGL(k_scatter_rows, dim3(grid_for(8 * N, DIF_BLOCK, 8192)), dim3(DIF_BLOCK), 0, s, (const uint32_t*)ws.pair_key, 8 * N,
                           (const int*)map->item_start, map->seg_start /* row cursors, idle 0 */, ws.row_val, ws.max_items * ITEM_ROWS);
    }
    DIF_CHECK_LAUNCH();
    {
        const size_t lds_bytes = (size_t)ENC_FLOATS * 4;
        static bool attr_set[64] = {};
        int dev = 0; (void)hipGetDevice(&dev);
        if (dev < 64 && !attr_set[dev]) {
            if (hipFuncSetAttribute((const void*)k_encode, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return DIF_ELAUNCH;
            attr_set[dev] = true;
        }
        ProfScope prof(DIF_PROF_ENCODE, s);
        hipLaunchKernelGGL(k_encode, dim3(num_cus()), dim3(512), lds_bytes, s, g, w->enc_packed, xyz, normal, N, (const uint32_t*)ws.row_val,
                           (const int*)map->seg_cnt, (const int*)map->item_start, (const int*)ws.item_slot, (const int*)C, ws.partial);
        DIF_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(k_fuse, dim3(grid_for(map->capacity * 32, DIF_BLOCK, 1024)), dim3(DIF_BLOCK), 0, s, (const long long*)ws.partial,
                       (const int*)map->item_start, map->seg_cnt, map->seg_start, map->latent_vecs, map->voxel_obs_count, map->dirty, C);
    DIF_CHECK_LAUNCH();
    return DIF_OK;
}

// ---- decoder launches ------------------------------------------------------------------------------------------
static int launch_decode(const DecodeArgs& A, const dif_weights_t* w, int64_t tiles_upper, hipStream_t s) {
    if (!w || !w->dec_packed || w->dec_packed_floats != DEC_FLOATS) return DIF_EINVAL;
    const bool grad = A.out_grad != nullptr;
    if (grad && (!w->dec_bwd_packed || w->dec_bwd_packed_floats != DECB_FLOATS)) return DIF_EINVAL;
    const size_t lds_bytes = (size_t)DEC_LDS_FLOATS * 4;
    static bool attr_set[64] = {};
    int dev = 0; (void)hipGetDevice(&dev);
    if (dev < 64 && !attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)k_decode<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return DIF_ELAUNCH;
        if (hipFuncSetAttribute((const void*)k_decode<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return DIF_ELAUNCH;
        attr_set[dev] = true;
    }
    int64_t blocks = (tiles_upper + 7) / 8;
    if (blocks < 1) blocks = 1;
    if (blocks > num_cus()) blocks = num_cus();
    ProfScope prof(A.mode == 0 ? DIF_PROF_DECODE_LATTICE : DIF_PROF_DECODE_POINTS, s);
    if (grad) {
        DecodeArgs B = A;
        B.wbwd = w->dec_bwd_packed;
        blocks = (tiles_upper + 3) / 4;
        if (blocks < 1) blocks = 1;
        if (blocks > num_cus()) blocks = num_cus();
        hipLaunchKernelGGL(k_decode<true>, dim3((int)blocks), dim3(256), lds_bytes, s, B, w->dec_packed);
    } else {
        hipLaunchKernelGGL(k_decode<false>, dim3((int)blocks), dim3(512), lds_bytes, s, A, w->dec_packed);
    }
    DIF_CHECK_LAUNCH();
    return DIF_OK;
}

int dif_decode_rows(const dif_weights_t* w, const float* rows, int64_t n, float* sdf, float* std_out, void* stream) {
    if (n < 0 || (n > 0 && (!rows || !sdf || !std_out))) return DIF_EINVAL;
    if (n == 0) return DIF_OK;
    DecodeArgs A = {};
    A.mode = 2; A.n_static = n; A.rows = rows; A.out_sdf = sdf; A.out_std = std_out; A.sign = 1.0f; A.lat.res = 1;
    return launch_decode(A, w, (n + 31) / 32, (hipStream_t)stream);
}

// encoder on explicit rows: one work item per 256 rows, partial sums are not what we want here, so a dedicated small kernel
namespace {
__global__ void __launch_bounds__(512, 2) k_encode_rows(const float* __restrict__ wblob, const float* __restrict__ rows, int64_t n, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    stage_weights(lds, wblob, ENC_FLOATS);
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
        f16v o = encoder_tile(lds, x0, x1, x2, lane);
        if (live) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int f = (r & 3) + 8 * (r >> 2) + 4 * half;
                if (f < L) out[row * L + f] = o[r];
            }
        }
    }
}
}  // namespace

int dif_encode_rows(const dif_weights_t* w, const float* rows, int64_t n, float* out, void* stream) {
    if (!w || !w->enc_packed || w->enc_packed_floats != ENC_FLOATS || n < 0 || (n > 0 && (!rows || !out))) return DIF_EINVAL;
    if (n == 0) return DIF_OK;
    const size_t lds_bytes = (size_t)ENC_FLOATS * 4;
    static bool attr_set[64] = {};
    int dev = 0; (void)hipGetDevice(&dev);
    if (dev < 64 && !attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)k_encode_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return DIF_ELAUNCH;
        attr_set[dev] = true;
    }
    int64_t blocks = ((n + 31) / 32 + 7) / 8;
    if (blocks > num_cus()) blocks = num_cus();
    hipLaunchKernelGGL(k_encode_rows, dim3((int)blocks), dim3(512), lds_bytes, (hipStream_t)stream, w->enc_packed, rows, n, out);
    DIF_CHECK_LAUNCH();
    return DIF_OK;
}

// ---- marching cubes --------------------------------------------------------------------------------------------
static int mc_setup(const McArgs& a, size_t& lds_bytes, int& blocks, int64_t K_upper) {
    if (upload_tables() != DIF_OK) return DIF_ELAUNCH;
    const int r = a.R / 2, nc = (r + 1) * (r + 1) * (r + 1);
    lds_bytes = (size_t)(DIF_BLOCK / 64) * (2 * nc + 32) * sizeof(float);
    if (lds_bytes > 64 * 1024) return DIF_EINVAL;
    blocks = grid_for(K_upper, DIF_BLOCK / 64, 8192);
    return DIF_OK;
}

static int mc_count_and_scan(McArgs a, int64_t K_upper, int32_t* tri_count, int32_t* tri_offset, int32_t* block_tmp, int* counters, hipStream_t s) {
    size_t lds_bytes; int blocks;
    int rc = mc_setup(a, lds_bytes, blocks, K_upper);
    if (rc != DIF_OK) return rc;
    a.tri_count = tri_count;
    a.tri_offset = tri_offset;
    {
        ProfScope prof(DIF_PROF_MC_COUNT, s);
        hipLaunchKernelGGL(k_marching_cubes<false>, dim3(blocks), dim3(DIF_BLOCK), lds_bytes, s, a);
    }
    DIF_CHECK_LAUNCH();
    TriScanFunctor f{tri_count, tri_offset, counters};
    return launch_scan(f, a.K_ptr, (int)a.K_static, K_upper, block_tmp, s);
}

static int mc_emit(McArgs a, int64_t K_upper, int32_t* tri_count, int32_t* tri_offset, hipStream_t s) {
    size_t lds_bytes; int blocks;
    int rc = mc_setup(a, lds_bytes, blocks, K_upper);
    if (rc != DIF_OK) return rc;
    a.tri_count = tri_count;
    a.tri_offset = tri_offset;
    {
        ProfScope prof(DIF_PROF_MC_EMIT, s);
        hipLaunchKernelGGL(k_marching_cubes<true>, dim3(blocks), dim3(DIF_BLOCK), lds_bytes, s, a);
    }
    DIF_CHECK_LAUNCH();
    return DIF_OK;
}

static int run_marching_cubes(McArgs a, int64_t K_upper, int32_t* tri_count, int32_t* tri_offset, int32_t* block_tmp, int* counters, hipStream_t s) {
    int rc = mc_count_and_scan(a, K_upper, tri_count, tri_offset, block_tmp, counters, s);
    if (rc != DIF_OK) return rc;
    return mc_emit(a, K_upper, tri_count, tri_offset, s);
}

int dif_marching_cubes(const int64_t* indexer, int32_t nx, int32_t ny, int32_t nz, const int64_t* valid_blocks, int64_t K,
                       const int32_t* vec_batch_mapping, int64_t V, const float* cube_sdf, const float* cube_std, int32_t R, float max_std,
                       int64_t max_triangles, float* triangles, int64_t* triangle_flatten_id, float* triangle_std, int32_t* tri_count,
                       int32_t* tri_offset, int32_t* block_tmp, int32_t* counters, void* stream) {
    if (!indexer || !counters || !block_tmp || K < 0 || V < 0 || R < 2 || (R & 1) || max_triangles < 0) return DIF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (K == 0) return hipMemsetAsync(counters + DIF_C_T, 0, sizeof(int), s) == hipSuccess ? DIF_OK : DIF_ELAUNCH;
    if (!valid_blocks || !vec_batch_mapping || !cube_sdf || !cube_std || !triangles || !triangle_flatten_id || !triangle_std || !tri_count || !tri_offset)
        return DIF_EINVAL;
    McArgs a = {};
    a.indexer = indexer; a.nx = nx; a.ny = ny; a.nz = nz; a.valid_blocks = valid_blocks; a.K_ptr = nullptr; a.K_static = K;
    a.vbm = vec_batch_mapping; a.V = V; a.cube_sdf = cube_sdf; a.cube_std = cube_std; a.R = R; a.max_std = max_std;
    a.max_triangles = max_triangles; a.new_limit = max_triangles; a.base_ptr = nullptr;
    a.triangles = triangles; a.tri_id = triangle_flatten_id; a.tri_std = triangle_std; a.tri_alive = nullptr; a.scale = 0;
    return run_marching_cubes(a, K, tri_count, tri_offset, block_tmp, counters, s);
}

// ---- extract ---------------------------------------------------------------------------------------------------
int dif_extract(const dif_map_t* map, const dif_weights_t* w, const dif_extract_buffers_t* buf, int32_t resolution, int32_t fast,
                float max_std, int32_t no_cache, int32_t scale_vertices, void* stream_) {
    if (!map || !w || !buf || resolution < 1 || resolution > 8 || buf->max_voxels <= 0) return DIF_EINVAL;
    if (buf->cache_capacity <= 0 || buf->cache_capacity >= ((int64_t)1 << 31) || !buf->cache_tri || !buf->cache_id || !buf->cache_std || !buf->cache_alive)
        return DIF_EINVAL;
    if (!map->tri_start || !map->tri_n) return DIF_EINVAL;
    hipStream_t s = (hipStream_t)stream_;
    const int64_t grid = (int64_t)map->nx * map->ny * map->nz;
    Geo g = geo_of(map);
    int* C = map->counters;
    const int r = resolution, R = 2 * r, l = r;               // fast two-level: low lattice l = R/2 (map.py:642-644)
    const int R3 = R * R * R;
    if (buf->max_voxels * (int64_t)R3 >= ((int64_t)1 << 31)) return DIF_EINVAL;
    const double sample_a = -(double)(r / 2) * (1.0 / r), sample_b = 1.0 + (double)((r - 1) / 2) * (1.0 / r);   // map.py:640-641

    {   // dirty slots -> valid_blocks
        const int64_t plane = (int64_t)map->ny * map->nz;
        const bool tiled = map->own_x_hi > map->own_x_lo && (map->own_x_lo > 0 || map->own_x_hi < map->nx);
        const int64_t own_lo = tiled ? map->own_x_lo * plane : 0, own_hi = tiled ? map->own_x_hi * plane : grid;
        if (tiled) {
            hipLaunchKernelGGL(k_mark_halo_dirty, dim3(grid_for(map->capacity, DIF_BLOCK, 256)), dim3(DIF_BLOCK), 0, s, g, map->ignore_count_th, map->dirty,
                               (const int64_t*)map->latent_vecs_pos, (const int64_t*)map->indexer, (const float*)map->voxel_obs_count, map->grid_bits,
                               (const int*)C, own_lo, own_hi);
            DIF_CHECK_LAUNCH();
        }
        DirtyFunctor f{map->dirty, map->latent_vecs_pos, buf->valid_blocks, C, no_cache, buf->max_voxels, g, map->ignore_count_th,
                       map->indexer, map->voxel_obs_count, map->grid_bits, own_lo, own_hi};
        if (launch_scan(f, C + DIF_C_N_OCCUPIED, 0, map->capacity, buf->block_tmp, s) != DIF_OK) return DIF_ELAUNCH;
    }
    {
        OccFunctor f{map->grid_bits, map->indexer, buf->occ_slot, map->vbm, C, buf->max_voxels};
        int nwords = (int)((grid + 31) / 32);
        if (launch_scan(f, nullptr, nwords, nwords, buf->block_tmp, s) != DIF_OK) return DIF_ELAUNCH;
    }
    int rc;
    if (fast && l * l * l <= VD_MAX_L3 && R * R <= VD_MAX_R2) {
        // fused per-voxel low lattice + upsample + threshold, then the balanced exact re-decode (map.py:644-679)
        if (!w->dec_packed || w->dec_packed_floats != DEC_FLOATS) return DIF_EINVAL;
        VoxelDecodeArgs V = {};
        V.occ_slot = buf->occ_slot; V.latent = map->latent_vecs; V.cube_sdf = buf->cube_sdf; V.cube_std = buf->cube_std; V.counters = C;
        V.refine_list = buf->refine_list; V.R = R;
        V.low.res = l; V.low.a = (float)sample_a; V.low.vsize = (l > 1) ? (float)((sample_b - sample_a) / (l - 1)) : 0.0f;
        const size_t lds_bytes = ((size_t)((DEC_LDS_FLOATS + 3) & ~3) + 8 * VD_WAVE_LDS_FLOATS) * 4;
        static bool attr_set[64] = {};
        int dev = 0; (void)hipGetDevice(&dev);
        if (dev < 64 && !attr_set[dev]) {
            if (hipFuncSetAttribute((const void*)k_decode_voxels, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return DIF_ELAUNCH;
            attr_set[dev] = true;
        }
        int64_t blocks = (buf->max_voxels + 7) / 8;
        if (blocks > num_cus()) blocks = num_cus();
        {
            ProfScope prof(DIF_PROF_DECODE_LATTICE, s);
            hipLaunchKernelGGL(k_decode_voxels, dim3((int)blocks), dim3(512), lds_bytes, s, V, w->dec_packed);
        }
        DIF_CHECK_LAUNCH();
        DecodeArgs Rf = {};
        Rf.mode = 1; Rf.n_ptr = C + DIF_C_VH; Rf.occ_slot = buf->occ_slot; Rf.latent = map->latent_vecs; Rf.list = buf->refine_list;
        Rf.lat.res = R; Rf.lat.a = (float)sample_a; Rf.lat.vsize = (float)((sample_b - sample_a) / (R - 1));
        Rf.out_sdf = buf->cube_sdf; Rf.out_std = buf->cube_std; Rf.sign = -1.0f;
        rc = launch_decode(Rf, w, buf->max_voxels * (int64_t)(R3 / 32), s);
        if (rc != DIF_OK) return rc;
    } else if (fast) {
        // low lattice decode (map.py:644-653)
        DecodeArgs A = {};
        A.mode = 0; A.n_ptr = C + DIF_C_B; A.occ_slot = buf->occ_slot; A.latent = map->latent_vecs;
        A.lat.res = l; A.lat.a = (float)sample_a;
        A.lat.vsize = (l > 1) ? (float)((sample_b - sample_a) / (l - 1)) : 0.0f;
        A.out_sdf = buf->low_sdf; A.out_std = buf->low_std; A.sign = 1.0f;
        rc = launch_decode(A, w, buf->max_voxels * ((l * l * l + 31) / 32), s);
        if (rc != DIF_OK) return rc;
        // upsample + threshold (map.py:655-667)
        hipLaunchKernelGGL(k_upsample_mark, dim3(grid_for(buf->max_voxels * (int64_t)(R * R), DIF_BLOCK, 4096)), dim3(DIF_BLOCK), 0, s,
                           (const float*)buf->low_sdf, (const float*)buf->low_std, l, R, buf->cube_sdf, buf->cube_std, buf->refine_list, C);
        DIF_CHECK_LAUNCH();
        // exact re-decode of the near-surface samples (map.py:668-679)
        DecodeArgs Rf = {};
        Rf.mode = 1; Rf.n_ptr = C + DIF_C_VH; Rf.occ_slot = buf->occ_slot; Rf.latent = map->latent_vecs; Rf.list = buf->refine_list;
        Rf.lat.res = R; Rf.lat.a = (float)sample_a; Rf.lat.vsize = (float)((sample_b - sample_a) / (R - 1));
        Rf.out_sdf = buf->cube_sdf; Rf.out_std = buf->cube_std; Rf.sign = -1.0f;
        rc = launch_decode(Rf, w, buf->max_voxels * (int64_t)(R3 / 32), s);
        if (rc != DIF_OK) return rc;
    } else {
        // every lattice sample decoded exactly (map.py:683-685), stored negated (map.py:687)
        DecodeArgs A = {};
        A.mode = 0; A.n_ptr = C + DIF_C_B; A.occ_slot = buf->occ_slot; A.latent = map->latent_vecs;
        A.lat.res = R; A.lat.a = (float)sample_a; A.lat.vsize = (float)((sample_b - sample_a) / (R - 1));
        A.out_sdf = buf->cube_sdf; A.out_std = buf->cube_std; A.sign = -1.0f;
        rc = launch_decode(A, w, buf->max_voxels * (int64_t)((R3 + 31) / 32), s);
        if (rc != DIF_OK) return rc;
    }
    // marching cubes (map.py:689-691)
    McArgs a = {};
    a.indexer = map->indexer; a.nx = map->nx; a.ny = map->ny; a.nz = map->nz; a.valid_blocks = buf->valid_blocks; a.K_ptr = C + DIF_C_K; a.K_static = 0;
    a.vbm = map->vbm; a.V = map->capacity; a.cube_sdf = buf->cube_sdf; a.cube_std = buf->cube_std; a.R = R; a.max_std = max_std;
    a.max_triangles = buf->cache_capacity; a.new_limit = buf->max_triangles; a.base_ptr = C + DIF_C_CACHE_T;     // append at the log's end
    a.triangles = buf->cache_tri; a.tri_id = buf->cache_id; a.tri_std = buf->cache_std; a.tri_alive = buf->cache_alive;
    a.scale = scale_vertices ? 1 : 0; a.vs = map->voxel_size; a.bx = map->bound_min[0]; a.by = map->bound_min[1]; a.bz = map->bound_min[2];
    if (no_cache) {                                                                                               // map.py:614-616
        if (hipMemsetAsync(C + DIF_C_CACHE_T, 0, sizeof(int), s) != hipSuccess) return DIF_ELAUNCH;
        if (hipMemsetAsync(C + DIF_C_CACHE_DEAD, 0, sizeof(int), s) != hipSuccess) return DIF_ELAUNCH;
        if (hipMemsetAsync(map->tri_n, 0, sizeof(int32_t) * (size_t)map->capacity, s) != hipSuccess) return DIF_ELAUNCH;
    }
    rc = mc_count_and_scan(a, buf->max_voxels, buf->tri_count, buf->tri_offset, buf->block_tmp, C, s);
    if (rc != DIF_OK) return rc;
    hipLaunchKernelGGL(k_log_replace, dim3(grid_for(buf->max_voxels, DIF_BLOCK, 256)), dim3(DIF_BLOCK), 0, s, (const int64_t*)buf->valid_blocks,
                       (const int32_t*)buf->tri_count, (const int32_t*)buf->tri_offset, (const int64_t*)map->indexer, map->tri_start, map->tri_n,
                       buf->cache_alive, C, buf->max_triangles, buf->cache_capacity);
    DIF_CHECK_LAUNCH();
    rc = mc_emit(a, buf->max_voxels, buf->tri_count, buf->tri_offset, s);
    if (rc != DIF_OK) return rc;
    hipLaunchKernelGGL(k_extract_finish, dim3(grid_for(buf->max_voxels, DIF_BLOCK, 256)), dim3(DIF_BLOCK), 0, s, (const int32_t*)buf->occ_slot, map->vbm,
                       C, buf->max_triangles, buf->cache_capacity);
    DIF_CHECK_LAUNCH();
    return DIF_OK;
}

int dif_mesh_cache_compact(const dif_map_t* map, const dif_extract_buffers_t* buf, float* out_tri, int64_t* out_id, float* out_std,
                           int64_t out_capacity, int32_t* scratch, void* stream) {
    if (!map || !buf || !out_tri || !out_id || !out_std || !scratch || out_capacity <= 0) return DIF_EINVAL;
    CacheLiveFunctor f{buf->cache_tri, buf->cache_id, buf->cache_std, buf->cache_alive, out_tri, out_id, out_std, out_capacity, map->counters};
    return launch_scan(f, map->counters + DIF_C_CACHE_T, 0, buf->cache_capacity, scratch, (hipStream_t)stream);
}

int dif_mesh_cache_reindex(const dif_map_t* map, const dif_extract_buffers_t* buf, int64_t n, void* stream) {
    if (!map || !buf || n < 0 || n > buf->cache_capacity) return DIF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(map->tri_n, 0, sizeof(int32_t) * (size_t)map->capacity, s) != hipSuccess) return DIF_ELAUNCH;
    hipLaunchKernelGGL(k_cache_reindex, dim3(grid_for(n > 0 ? n : 1)), dim3(DIF_BLOCK), 0, s, (const int64_t*)buf->cache_id, n, (const int64_t*)map->indexer,
                       map->tri_start, map->tri_n, buf->cache_alive, map->counters);
    DIF_CHECK_LAUNCH();
    return DIF_OK;
}

// ---- get_sdf ---------------------------------------------------------------------------------------------------
int dif_query_sdf(const dif_map_t* map, const dif_weights_t* w, const float* xyz, int64_t N, uint8_t* mask, int32_t* sel, float* sdf,
                  float* std_out, float* grad, int32_t* scratch, void* stream_) {
    if (!map || !w || N < 0 || N >= ((int64_t)1 << 31)) return DIF_EINVAL;
    hipStream_t s = (hipStream_t)stream_;
    if (N == 0) return hipMemsetAsync(map->counters + DIF_C_QUERY_M, 0, sizeof(int), s) == hipSuccess ? DIF_OK : DIF_ELAUNCH;
    if (!xyz || !mask || !sel || !sdf || !std_out || !scratch) return DIF_EINVAL;
    Geo g = geo_of(map);
    QueryFunctor f{g, map->ignore_count_th, xyz, map->indexer, map->voxel_obs_count, mask, sel, map->counters};
    if (launch_scan(f, nullptr, (int)N, N, scratch, s) != DIF_OK) return DIF_ELAUNCH;
    DecodeArgs A = {};
    A.mode = 3; A.n_ptr = map->counters + DIF_C_QUERY_M; A.latent = map->latent_vecs; A.list = sel; A.xyz = xyz; A.indexer = map->indexer;
    A.geo = g; A.out_sdf = sdf; A.out_std = std_out; A.sign = 1.0f; A.lat.res = 1;
    A.out_grad = grad; A.grad_scale = 1.0f / map->voxel_size;
    return launch_decode(A, w, (N + 31) / 32, s);
}

// ---- multi-GPU merge -------------------------------------------------------------------------------------------
int dif_export_records(const dif_map_t* map, int32_t* records, int64_t max_records, int32_t x_lo, int32_t x_hi, int32_t raw, int32_t* scratch,
                       void* stream) {
    if (!map || !records || !scratch || max_records <= 0) return DIF_EINVAL;
    if (x_lo < 0) x_lo = 0;
    if (x_hi > map->nx) x_hi = map->nx;
    const int64_t plane = (int64_t)map->ny * map->nz;
    ExportFunctor f{map->latent_vecs_pos, map->voxel_obs_count, map->latent_vecs, map->dirty, records, max_records, x_lo * plane, (x_hi > x_lo ? x_hi : x_lo) * plane,
                    raw ? 1 : 0, map->counters};
    return launch_scan(f, map->counters + DIF_C_N_OCCUPIED, 0, map->capacity, scratch, (hipStream_t)stream);
}

int dif_merge_records(const dif_map_t* map, const int32_t* records, int64_t n, int32_t assign, int32_t* scratch, void* stream_) {
    if (!map || n < 0) return DIF_EINVAL;
    if (n == 0) return DIF_OK;
    if (!records || !scratch) return DIF_EINVAL;
    hipStream_t s = (hipStream_t)stream_;
    const int64_t grid = (int64_t)map->nx * map->ny * map->nz;
    if (hipMemsetAsync(map->counters + DIF_C_ALLOC_NEW, 0, sizeof(int), s) != hipSuccess) return DIF_ELAUNCH;
    hipLaunchKernelGGL(k_merge_mark, dim3(grid_for(n)), dim3(DIF_BLOCK), 0, s, records, n, (const int64_t*)map->indexer, map->grid_bits, grid);
    DIF_CHECK_LAUNCH();
    AllocFunctor f{map->grid_bits, map->indexer, map->latent_vecs_pos, map->counters, map->capacity};
    int nwords = (int)((grid + 31) / 32);
    if (launch_scan(f, nullptr, nwords, nwords, scratch, s) != DIF_OK) return DIF_ELAUNCH;
    hipLaunchKernelGGL(k_merge_apply, dim3(grid_for(n * 32, DIF_BLOCK, 2048)), dim3(DIF_BLOCK), 0, s, records, n, (const int64_t*)map->indexer,
                       map->latent_vecs, map->voxel_obs_count, map->dirty, map->counters, grid, map->capacity, assign ? 1 : 0);
    DIF_CHECK_LAUNCH();
    return DIF_OK;
}

int dif_profile_enable(int32_t on) {
    g_prof_on = on != 0;
    return DIF_OK;
}

int dif_profile_read(double* ms, int64_t* launches, int32_t reset) {
    if (!ms || !launches) return DIF_EINVAL;
    for (int i = 0; i < DIF_PROF_COUNT; ++i) { ms[i] = 0.0; launches[i] = 0; }
    for (auto& r : g_prof) {
        if (hipEventSynchronize(r.b) != hipSuccess) return DIF_ELAUNCH;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return DIF_ELAUNCH;
        ms[r.which] += t;
        launches[r.which] += 1;
    }
    if (reset) {
        for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        g_prof.clear();
    }
    return DIF_OK;
}

int dif_read_counters(const dif_map_t* map, int32_t* host_out, void* stream) {
    if (!map || !host_out) return DIF_EINVAL;
    if (hipMemcpyAsync(host_out, map->counters, sizeof(int32_t) * DIF_C_COUNT, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return DIF_ELAUNCH;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return DIF_ELAUNCH;
    return DIF_OK;
}

}  // extern "C"
