// Image / point-cloud kernels on either side of the path: a1-a2 depth -> points, normals, 8f-2 preprocessing, flat groupby_sum  (part of libdifusion; included by difusion.hip inside its anonymous namespace)
#pragma once

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

// One pixel: back-projection (imgproc.cu:18-20 op order) and pose transform ((r0*x + r1*y) + r2*z) + t, every op rounded
// (synthetic.transform_points states the same order).  NaN depth -> NaN point and normal.
__device__ __forceinline__ void unproject_point(const float* __restrict__ depth, const float* __restrict__ ncam, int64_t i, int W, float fx, float fy,
                                                float cx, float cy, const Pose& P, float (&p)[3], float (&nv)[3]) {
    const int v = (int)(i / W), u = (int)(i - (int64_t)v * W);
    const float d = depth[i];
    const float qnan = __builtin_nanf("");
    p[0] = p[1] = p[2] = nv[0] = nv[1] = nv[2] = qnan;
    if (d == d) {
        const float x = ((float)u - cx) / fx * d;
        const float y = ((float)v - cy) / fy * d;
        const float z = d;
        p[0] = ((P.r[0] * x + P.r[1] * y) + P.r[2] * z) + P.t[0];
        p[1] = ((P.r[3] * x + P.r[4] * y) + P.r[5] * z) + P.t[1];
        p[2] = ((P.r[6] * x + P.r[7] * y) + P.r[8] * z) + P.t[2];
        if (ncam) {
            const float a = ncam[i * 3 + 0], b = ncam[i * 3 + 1], c = ncam[i * 3 + 2];
            nv[0] = (P.r[0] * a + P.r[1] * b) + P.r[2] * c;
            nv[1] = (P.r[3] * a + P.r[4] * b) + P.r[5] * c;
            nv[2] = (P.r[6] * a + P.r[7] * b) + P.r[8] * c;
        }
    }
}

// Deferred export (dif_map_t.pending_export): workgroup `part` of `nparts` copies its share of the previous extract's new triangles to the
// caller's arrays (pinned host memory: PCIe-bound, ~0.5 MB per frame) — run by a few extra workgroups of the next frame's first kernel.
__device__ __forceinline__ void copy_dwords_strided(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int64_t n, int64_t t0, int64_t stride) {
    int64_t j = t0;
    for (; j + 3 * stride < n; j += 4 * stride) {           // four loads in flight per thread, then four stores: the copy is latency-bound
        const uint32_t a = src[j], b = src[j + stride], c = src[j + 2 * stride], d = src[j + 3 * stride];
        dst[j] = a; dst[j + stride] = b; dst[j + 2 * stride] = c; dst[j + 3 * stride] = d;
    }
    for (; j < n; j += stride) dst[j] = src[j];
}

__device__ __forceinline__ void export_pending_rows(const dif_pending_export_t* __restrict__ d, int part, int nparts) {
    if (!d->pending) return;
    const int64_t kept = d->kept, n = d->n;
    const int64_t stride = (int64_t)nparts * blockDim.x, t0 = (int64_t)part * blockDim.x + threadIdx.x;
    // consecutive lanes write consecutive dwords: 256 contiguous bytes per wave and store instruction (whole PCIe write bursts)
    copy_dwords_strided(reinterpret_cast<const uint32_t*>(d->log_tri + kept * 9), reinterpret_cast<uint32_t*>(d->out_tri), n * 9, t0, stride);
    copy_dwords_strided(reinterpret_cast<const uint32_t*>(d->log_std + kept * 3), reinterpret_cast<uint32_t*>(d->out_std), n * 3, t0, stride);
    copy_dwords_strided(reinterpret_cast<const uint32_t*>(d->log_id + kept), reinterpret_cast<uint32_t*>(d->out_id), n * 2, t0, stride);
}

#define DIF_EXPORT_WGS 64

__global__ void __launch_bounds__(DIF_BLOCK) k_export_pending(dif_pending_export_t* __restrict__ d) {
    export_pending_rows(d, (int)blockIdx.x, (int)gridDim.x);
}

__global__ void __launch_bounds__(DIF_BLOCK) k_unproject_transform(const float* __restrict__ depth, const float* __restrict__ ncam,
                                                                 float* __restrict__ xyz, float* __restrict__ nrm, int H, int W,
                                                                 float fx, float fy, float cx, float cy, Pose P, const float* __restrict__ pose_dev,
                                                                 const dif_frame_t* __restrict__ frame) {
    if (frame) {                                      // whole frame descriptor in device memory: a captured hipGraph replays on new inputs
        depth = frame->depth;                         // after one 64-byte upload, without staging copies
        ncam = frame->normal_cam;
        pose_dev = frame->pose;
    }
    if (pose_dev) {                                   // pose read from device memory: lets a captured hipGraph be replayed per frame
#pragma unroll
        for (int i = 0; i < 9; ++i) P.r[i] = pose_dev[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) P.t[i] = pose_dev[9 + i];
    }
    int64_t n = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float p[3], nv[3];
        unproject_point(depth, ncam, i, W, fx, fy, cx, cy, P, p, nv);
        xyz[i * 3 + 0] = p[0]; xyz[i * 3 + 1] = p[1]; xyz[i * 3 + 2] = p[2];
        if (nrm) { nrm[i * 3 + 0] = nv[0]; nrm[i * 3 + 1] = nv[1]; nrm[i * 3 + 2] = nv[2]; }
    }
}

// ---- 8f-2: image-space preprocessing in front of the path -----------------------------------------------------------------
// The reference runs three one-thread-per-pixel kernels with a round trip through memory between them: filter_depth (5x5
// bilateral filter, imgproc.cu:48-94), unproject_depth (:5-44) and compute_normal_weight (central-difference normal + noise-model
// weight, :98-160).  Here ONE workgroup owns a 16 x 16 pixel tile: the raw depth tile with its 3-pixel apron (22 x 22) goes into LDS
// once, the filtered depth of the tile plus a 1-pixel apron (18 x 18) is computed from it into LDS, and every thread then
// back-projects its pixel and its four neighbours out of LDS and writes depth / point / normal+weight exactly once.  The
// arithmetic of every stage is the reference's, operation for operation (the library is built without FMA contraction), so the
// outputs equal the three-kernel composition bit for bit.
struct V3 { float x, y, z; };

__device__ __forceinline__ V3 backproject(int u, int v, float d, float fx, float fy, float cx, float cy) {      // imgproc.cu:18-20
    const float qnan = __builtin_nanf("");
    if (!(d == d)) return V3{qnan, qnan, qnan};
    return V3{((float)u - cx) / fx * d, ((float)v - cy) / fy * d, d};
}

// imgproc.cu:105-141 for an interior pixel: false => weight -1
__device__ __forceinline__ bool normal_weight_of(const V3& c, const V3& xp, const V3& xm, const V3& yp, const V3& ym, float (&o)[4]) {
    if (c.z <= 1e-6) return false;
    if (xp.z < 1e-6 || xm.z < 1e-6 || yp.z < 1e-6 || ym.z < 1e-6) return false;
    const float dxx = xp.x - xm.x, dxy = xp.y - xm.y, dxz = xp.z - xm.z;
    const float dyx = yp.x - ym.x, dyy = yp.y - ym.y, dyz = yp.z - ym.z;
    float nx = dyy * dxz - dyz * dxy, ny = dyz * dxx - dyx * dxz, nz = dyx * dxy - dyy * dxx;      // cross(diff_y, diff_x)
    const float len = sqrtf(nx * nx + ny * ny + nz * nz);
    if (len < 1e-6) return false;
    nx /= len; ny /= len; nz /= len;
    const float theta = acosf(nz);
    const float td = theta / (0.5f * 3.14159f - theta);
    const float wgt = (0.0012f + 0.0019f * (c.z - 0.4f) * (c.z - 0.4f) + 0.0001f / sqrtf(c.z) * td * td);
    o[0] = nx; o[1] = ny; o[2] = nz; o[3] = 1.0f / wgt;
    return true;
}

#define FE_TILE 16
#define FE_RAW (FE_TILE + 6)        /* 5x5 filter (2) behind a 4-neighbour stencil (1): 3-pixel apron */
#define FE_FIL (FE_TILE + 2)

struct FrontendArgs {
    const float* depth; int H, W;
    float fx, fy, cx, cy;
    int filter;                      // 0: the stencil runs on the raw depth
    float* depth_out; int write_border;      // filtered depth; the 2-pixel image border carries the raw depth (written only if write_border)
    float* pc;                       // (H,W,3) back-projected (filtered) depth
    float* normal_weight;            // (H,W,4)
    float* frame_depth;              // (H,W): filtered depth where a normal exists, NaN elsewhere  \  what dif_frame_t wants: a depth-only
    float* frame_normal;             // (H,W,3): that normal, NaN elsewhere                          /  stream can be integrated directly
};

__global__ void __launch_bounds__(FE_TILE * FE_TILE) k_depth_frontend(FrontendArgs a) {
    __shared__ float raw[FE_RAW][FE_RAW + 1];
    __shared__ float fil[FE_FIL][FE_FIL + 1];
    const int u0 = (int)blockIdx.x * FE_TILE, v0 = (int)blockIdx.y * FE_TILE, t = (int)threadIdx.x;
    const float qnan = __builtin_nanf("");
    for (int k = t; k < FE_RAW * FE_RAW; k += FE_TILE * FE_TILE) {
        const int i = k / FE_RAW, j = k % FE_RAW, v = v0 - 3 + i, u = u0 - 3 + j;
        raw[i][j] = (v >= 0 && v < a.H && u >= 0 && u < a.W) ? a.depth[(int64_t)v * a.W + u] : qnan;
    }
    __syncthreads();
    const float sig_l2 = 1.2232f * 1.2232f;                  // MEAN_SIGMA_L^2
    for (int k = t; k < FE_FIL * FE_FIL; k += FE_TILE * FE_TILE) {
        const int i = k / FE_FIL, j = k % FE_FIL, v = v0 - 1 + i, u = u0 - 1 + j;
        float z = raw[i + 2][j + 2];
        if (a.filter && v >= 2 && v < a.H - 2 && u >= 2 && u < a.W - 2) {      // imgproc.cu:52-79; border pixels keep the raw depth
            if (z < 1e-6) {
                z = 0.0f;
            } else {
                const float sigma_z = 1.0f / (0.0012f + 0.0019f * (z - 0.4f) * (z - 0.4f) + 0.0001f / sqrtf(z) * 0.25f);
                float w_sum = 0.0f, acc = 0.0f;
                for (int di = -2; di <= 2; ++di)
                    for (int dj = -2; dj <= 2; ++dj) {
                        const float nz = raw[i + 2 + di][j + 2 + dj];
                        if (nz < 1e-6) continue;
                        const float dz = (nz - z) * (nz - z);
                        const float wgt = expf(-0.5f * ((float)(abs(di) + abs(dj)) * sig_l2 + dz * sigma_z * sigma_z));
                        w_sum += wgt;
                        acc += wgt * nz;
                    }
                z = acc / w_sum;
            }
        }
        fil[i][j] = z;
    }
    __syncthreads();
    const int lx = t & (FE_TILE - 1), ly = t / FE_TILE, u = u0 + lx, v = v0 + ly;
    if (u >= a.W || v >= a.H) return;
    const int64_t px = (int64_t)v * a.W + u;
    const float d = fil[ly + 1][lx + 1];
    const bool interior = v >= 2 && v < a.H - 2 && u >= 2 && u < a.W - 2;
    if (a.depth_out && (interior || a.write_border)) a.depth_out[px] = d;
    const V3 c = backproject(u, v, d, a.fx, a.fy, a.cx, a.cy);
    if (a.pc) { a.pc[px * 3 + 0] = c.x; a.pc[px * 3 + 1] = c.y; a.pc[px * 3 + 2] = c.z; }
    if (!a.normal_weight && !a.frame_normal) return;
    float o[4] = {0.0f, 0.0f, 0.0f, -1.0f};
    bool ok = false;
    if (!(v < 1 || v > a.H - 2 || u < 1 || u > a.W - 2))                       // imgproc.cu:102
        ok = normal_weight_of(c, backproject(u + 1, v, fil[ly + 1][lx + 2], a.fx, a.fy, a.cx, a.cy),
                              backproject(u - 1, v, fil[ly + 1][lx], a.fx, a.fy, a.cx, a.cy),
                              backproject(u, v + 1, fil[ly + 2][lx + 1], a.fx, a.fy, a.cx, a.cy),
                              backproject(u, v - 1, fil[ly][lx + 1], a.fx, a.fy, a.cx, a.cy), o);
    if (a.normal_weight) {
        if (ok) { a.normal_weight[px * 4 + 0] = o[0]; a.normal_weight[px * 4 + 1] = o[1]; a.normal_weight[px * 4 + 2] = o[2]; }
        a.normal_weight[px * 4 + 3] = ok ? o[3] : -1.0f;                       // the reference writes only the weight of an invalid pixel
    }
    if (a.frame_normal) {
        const bool use = ok && o[3] > 0.0f && (o[0] == o[0]) && (d == d);      // a finite normal with a valid weight on a finite depth
        a.frame_depth[px] = use ? d : qnan;
        a.frame_normal[px * 3 + 0] = use ? o[0] : qnan; a.frame_normal[px * 3 + 1] = use ? o[1] : qnan; a.frame_normal[px * 3 + 2] = use ? o[2] : qnan;
    }
}

// compute_normal_weight on an arbitrary point map (the flat op of ext/__init__.py:26): the same stencil with the point tile and its
// 1-pixel apron staged through LDS (each point is read once instead of up to five times).
__global__ void __launch_bounds__(FE_TILE * FE_TILE) k_normal_weight(const float* __restrict__ pc, float* __restrict__ out, int H, int W) {
    __shared__ float tile[FE_FIL][FE_FIL][3];
    const int u0 = (int)blockIdx.x * FE_TILE, v0 = (int)blockIdx.y * FE_TILE, t = (int)threadIdx.x;
    for (int k = t; k < FE_FIL * FE_FIL; k += FE_TILE * FE_TILE) {
        const int i = k / FE_FIL, j = k % FE_FIL, v = v0 - 1 + i, u = u0 - 1 + j;
        const bool in = v >= 0 && v < H && u >= 0 && u < W;
        const float* p = pc + ((int64_t)v * W + u) * 3;
        tile[i][j][0] = in ? p[0] : 0.0f; tile[i][j][1] = in ? p[1] : 0.0f; tile[i][j][2] = in ? p[2] : 0.0f;
    }
    __syncthreads();
    const int lx = t & (FE_TILE - 1), ly = t / FE_TILE, u = u0 + lx, v = v0 + ly;
    if (u >= W || v >= H) return;
    float* o4 = out + ((int64_t)v * W + u) * 4;
    float o[4];
    bool ok = false;
    if (!(v < 1 || v > H - 2 || u < 1 || u > W - 2)) {
        auto at = [&](int i, int j) { return V3{tile[i][j][0], tile[i][j][1], tile[i][j][2]}; };
        ok = normal_weight_of(at(ly + 1, lx + 1), at(ly + 1, lx + 2), at(ly + 1, lx), at(ly + 2, lx + 1), at(ly, lx + 1), o);
    }
    if (ok) { o4[0] = o[0]; o4[1] = o[1]; o4[2] = o[2]; }
    o4[3] = ok ? o[3] : -1.0f;
}

// point_box_filter (system/tracker.py:13-23): mean point / mean normal per voxel_size box, boxes in ascending linear id
// (x fastest).  Bounds -> box bitmap -> ordered ranks -> order-independent fixed-point sums -> means.
struct BoxGrid { float minb[3]; int n[3]; };

__device__ __forceinline__ unsigned f2ord(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

// (a few dozen workgroups, one set of six atomics each: a wave-level atomic per 64 points queues ~6,000 of them on six addresses — 75 us
// for the tracker's 67 k points)
__global__ void __launch_bounds__(DIF_BLOCK) k_pbf_bounds(const float* __restrict__ pts, int64_t N, unsigned* __restrict__ mm /* [6]: min xyz, max xyz (ordered uint) */) {
    __shared__ unsigned s_lo[3][DIF_BLOCK / 64], s_hi[3][DIF_BLOCK / 64];
    unsigned lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x)
#pragma unroll
        for (int a = 0; a < 3; ++a) { unsigned o = f2ord(pts[i * 3 + a]); lo[a] = min(lo[a], o); hi[a] = max(hi[a], o); }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int d = 32; d >= 1; d >>= 1) { lo[a] = min(lo[a], (unsigned)__shfl_xor((int)lo[a], d)); hi[a] = max(hi[a], (unsigned)__shfl_xor((int)hi[a], d)); }
        if (lane_id() == 0) { s_lo[a][threadIdx.x >> 6] = lo[a]; s_hi[a][threadIdx.x >> 6] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = (int)threadIdx.x;
        unsigned l = s_lo[a][0], h = s_hi[a][0];
        for (int w = 1; w < DIF_BLOCK / 64; ++w) { l = min(l, s_lo[a][w]); h = max(h, s_hi[a][w]); }
        atomicMin(mm + a, l);
        atomicMax(mm + 3 + a, h);
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
    const int64_t cells = (int64_t)G.n[0] * G.n[1] * G.n[2];
    if (blockIdx.x == 0 && threadIdx.x == 0) status[1] = cells > max_cells ? 0 : (int)((cells + 31) >> 5);      // bitmap words in use: all the rank scan walks
    if (cells > max_cells) { if (blockIdx.x == 0 && threadIdx.x == 0) status[0] = 1; return; }
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
    NO_PACKED_F32
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < (int64_t)nb * 3; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = e / 3; int a = (int)(e - r * 3);
        const float cnt = (float)sums[r * 8 + 6];
        out_pts[e] = (float)((double)sums[r * 8 + a] * (1.0 / 16777216.0)) / cnt;
        out_nrm[e] = (float)((double)sums[r * 8 + 3 + a] * (1.0 / 16777216.0)) / cnt;
    }
    if (status[0]) return;
    const BoxGrid G = pbf_grid(mm, vs);                      // restore the bitmap to all-zero for the next call
    NO_PACKED_F32
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

