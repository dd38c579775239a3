// Point-cloud neighbourhood ops of the tracker's pre-processing: radius outlier removal and PCA normals
// over the k nearest neighbours.  Reference: ext/pcproc/pcproc.cu:98-209 on top of a FLANN-derived CUDA kd-tree
// (cuda_kdtree.cu:644-1260).  Here: a uniform grid of cell size c hashed into an open-addressing table (no bounding box, no
// sort) and an exact ring-by-ring search that stops as soon as the k-th distance is below the lower bound of everything not
// yet visited, or that bound passes the search radius.  Results are the exact kNN under the total order (d2, index), so
// they do not depend on the (racy) order in which points land inside a cell.
#pragma once

#define CLOUD_EMPTY 0xFFFFFFFFFFFFFFFFull
#define CLOUD_OFF (1 << 20)

struct CloudGrid {
    unsigned long long* keys;   // [T] cell key or CLOUD_EMPTY
    int* cnt;                   // [T] points in the cell
    int* end;                   // [T] after placement: one past the cell's last row in `sorted`
    float4* sorted;             // [n] (x, y, z, original index as int bits), grouped by cell
    int* slot;                  // [n] table slot of point i, -1 = not a finite point
    unsigned mask;              // T - 1
    int shift;                  // 64 - log2(T)
    float c, inv_c;
};

__device__ __forceinline__ bool cloud_cell(const CloudGrid& g, float x, float y, float z, int& cx, int& cy, int& cz) {
    float fx = floorf(x * g.inv_c), fy = floorf(y * g.inv_c), fz = floorf(z * g.inv_c);
    const float lim = (float)(CLOUD_OFF - 64);
    bool ok = (fabsf(fx) < lim) && (fabsf(fy) < lim) && (fabsf(fz) < lim);   // NaN / inf / far away fail
    cx = ok ? (int)fx : 0;
    cy = ok ? (int)fy : 0;
    cz = ok ? (int)fz : 0;
    return ok;
}

__device__ __forceinline__ unsigned long long cloud_key(int cx, int cy, int cz) {
    return ((unsigned long long)(unsigned)(cx + CLOUD_OFF) << 42) | ((unsigned long long)(unsigned)(cy + CLOUD_OFF) << 21) |
           (unsigned long long)(unsigned)(cz + CLOUD_OFF);
}

__device__ __forceinline__ unsigned cloud_hash(const CloudGrid& g, unsigned long long key) {
    return (unsigned)((key * 0x9E3779B97F4A7C15ull) >> g.shift);
}

__device__ __forceinline__ int cloud_find(const CloudGrid& g, unsigned long long key) {
    unsigned h = cloud_hash(g, key);
    while (true) {
        unsigned long long k = g.keys[h];
        if (k == key) return (int)h;
        if (k == CLOUD_EMPTY) return -1;
        h = (h + 1) & g.mask;
    }
}

__global__ void __launch_bounds__(DIF_BLOCK) k_cloud_insert(CloudGrid g, const float* __restrict__ pc, int n, int stride) {
    for (int i = blockIdx.x * DIF_BLOCK + threadIdx.x; i < n; i += gridDim.x * DIF_BLOCK) {
        const float* p = pc + (size_t)i * stride;
        int cx, cy, cz;
        int slot = -1;
        if (cloud_cell(g, p[0], p[1], p[2], cx, cy, cz)) {
            unsigned long long key = cloud_key(cx, cy, cz);
            unsigned h = cloud_hash(g, key);
            while (true) {
                unsigned long long prev = atomicCAS(&g.keys[h], CLOUD_EMPTY, key);
                if (prev == CLOUD_EMPTY || prev == key) break;
                h = (h + 1) & g.mask;
            }
            slot = (int)h;
            atomicAdd(&g.cnt[h], 1);
        }
        g.slot[i] = slot;
    }
}

struct CloudStartFunctor {
    const int* cnt;
    int* end;
    __device__ int count(int i) const { return cnt[i]; }
    __device__ void emit(int i, int off) const { end[i] = off; }     // becomes the placement cursor
    __device__ void finish(int) const {}
};

__global__ void __launch_bounds__(DIF_BLOCK) k_cloud_place(CloudGrid g, const float* __restrict__ pc, int n, int stride) {
    for (int i = blockIdx.x * DIF_BLOCK + threadIdx.x; i < n; i += gridDim.x * DIF_BLOCK) {
        int slot = g.slot[i];
        if (slot < 0) continue;
        const float* p = pc + (size_t)i * stride;
        int pos = atomicAdd(&g.end[slot], 1);
        g.sorted[pos] = make_float4(p[0], p[1], p[2], __int_as_float(i));
    }
}

// ---- bounded top-K under the total order (d2, index) -------------------------------------------------------------------
template <int K>
struct TopK {
    // vector-typed so the list lives in VGPRs whatever the unroller decides (a plain array of 16 landed in scratch)
    float __attribute__((ext_vector_type(K))) d;
    int __attribute__((ext_vector_type(K))) id;
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < K; ++j) { d[j] = __builtin_inff(); id[j] = 0x7FFFFFFF; }
    }
    __device__ __forceinline__ void push(float dist, int idx) {
        if (!(dist < d[K - 1] || (dist == d[K - 1] && idx < id[K - 1]))) return;
        float cd = dist;
        int ci = idx;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            bool lt = cd < d[j] || (cd == d[j] && ci < id[j]);
            float td = d[j];
            int ti = id[j];
            d[j] = lt ? cd : td;
            id[j] = lt ? ci : ti;
            cd = lt ? td : cd;
            ci = lt ? ti : ci;
        }
    }
    __device__ __forceinline__ float kth(int k) const {   // d[k-1] without dynamic register indexing
        float v = d[K - 1];
#pragma unroll
        for (int j = 0; j < K - 1; ++j) v = (j == k - 1) ? d[j] : v;
        return v;
    }
};

// pcproc.cu:21-96 (eigenvector of the smallest eigenvalue of a symmetric 3x3 matrix, trigonometric closed form).  The phase
// shift is added and its cosine taken in double precision there (M_PI is a double constant), mirrored here.
__device__ inline void sym3eig_min(float a11, float a12, float a13, float a21, float a22, float a23, float a31, float a32, float a33,
                                   float& nx, float& ny, float& nz) {
    const float p1 = a12 * a12 + a13 * a13 + a23 * a23;
    const float q = (a11 + a22 + a33) / 3.0f;
    const float p2 = (a11 - q) * (a11 - q) + (a22 - q) * (a22 - q) + (a33 - q) * (a33 - q) + 2 * p1;
    const float p = sqrtf(p2 / 6.0f);
    const float ip = 1.0f / p;
    const float b11 = ip * (a11 - q), b12 = ip * a12, b13 = ip * a13;
    const float b21 = ip * a21, b22 = ip * (a22 - q), b23 = ip * a23;
    const float b31 = ip * a31, b32 = ip * a32, b33 = ip * (a33 - q);
    float r = b11 * b22 * b33 + b12 * b23 * b31 + b13 * b21 * b32 - b13 * b22 * b31 - b12 * b21 * b33 - b11 * b23 * b32;
    r = r / 2.0f;
    float phi;
    if (r <= -1) phi = (float)(3.14159265358979323846 / 3.0);
    else if (r >= 1) phi = 0;
    else phi = acosf(r) / 3.0f;
    const float ev = (float)((double)q + (double)(2 * p) * cos((double)phi + (2 * 3.14159265358979323846 / 3)));
    a11 -= ev; a22 -= ev; a33 -= ev;
    const float r12_1 = a12 * a23 - a13 * a22, r12_2 = a13 * a21 - a11 * a23, r12_3 = a11 * a22 - a12 * a21;
    const float r13_1 = a12 * a33 - a13 * a32, r13_2 = a13 * a31 - a11 * a33, r13_3 = a11 * a32 - a12 * a31;
    const float r23_1 = a22 * a33 - a23 * a32, r23_2 = a23 * a31 - a21 * a33, r23_3 = a21 * a32 - a22 * a31;
    const float d1 = r12_1 * r12_1 + r12_2 * r12_2 + r12_3 * r12_3;
    const float d2 = r13_1 * r13_1 + r13_2 * r13_2 + r13_3 * r13_3;
    const float d3 = r23_1 * r23_1 + r23_2 * r23_2 + r23_3 * r23_3;
    float d_max = d1;
    int i_max = 0;
    if (d2 > d_max) { d_max = d2; i_max = 1; }
    if (d3 > d_max) i_max = 2;
    if (i_max == 0) { float s = sqrtf(d1); nx = r12_1 / s; ny = r12_2 / s; nz = r12_3 / s; }
    else if (i_max == 1) { float s = sqrtf(d2); nx = r13_1 / s; ny = r13_2 / s; nz = r13_3 / s; }
    else { float s = sqrtf(d3); nx = r23_1 / s; ny = r23_2 / s; nz = r23_3 / s; }
}

enum { CLOUD_KNN = 0, CLOUD_OUTLIER = 1, CLOUD_NORMAL = 2 };

struct CloudQueryOut {
    int* idx;            // KNN: (n, k)
    float* dist;         // KNN: (n, k)
    uint8_t* mask;       // OUTLIER: (n)
    float* normal;       // NORMAL: (n, 3)
    float cam[3];
};

// One thread per point, in cell order (so a wave's lanes walk the same few cells).  Ring rho = the shell of cells at Chebyshev
// distance rho from the query's cell; once rings 0..rho are done every unvisited point is farther than rho*c.
template <int K, int MODE>
__global__ void __launch_bounds__(DIF_BLOCK) k_cloud_query(CloudGrid g, const float* __restrict__ pc, int n, int stride, int k, float radius,
                                                           int max_ring, CloudQueryOut out) {
    const int row = blockIdx.x * DIF_BLOCK + threadIdx.x;
    // rows of `sorted` beyond the finite points do not exist; the finite count is the table's total
    if (row >= n) return;
    // non-finite points never got a row: they are handled by k_cloud_invalid
    const float4 q = g.sorted[row];
    const int qi = __float_as_int(q.w);
    if (qi < 0) return;                                    // unused tail row (sorted is pre-filled with index -1)
    int cx, cy, cz;
    cloud_cell(g, q.x, q.y, q.z, cx, cy, cz);
    const float r2 = radius * radius;
    TopK<K> top;
    top.init();
    for (int rho = 0; rho <= max_ring; ++rho) {
        for (int dx = -rho; dx <= rho; ++dx) {
            for (int dy = -rho; dy <= rho; ++dy) {
                const bool face = (dx == -rho) || (dx == rho) || (dy == -rho) || (dy == rho);
                const int step = (face || rho == 0) ? 1 : 2 * rho;      // interior columns: only the two end caps
                for (int dz = -rho; dz <= rho; dz += step) {
                    int s = cloud_find(g, cloud_key(cx + dx, cy + dy, cz + dz));
                    if (s < 0) continue;
                    const int e = g.end[s], b = e - g.cnt[s];
                    for (int j = b; j < e; ++j) {
                        const float4 p = g.sorted[j];
                        const float ex = p.x - q.x, ey = p.y - q.y, ez = p.z - q.z;
                        const float d2 = (ex * ex + ey * ey) + ez * ez;       // CudaL2::dist (cuda_kdtree.cu:1152-1155)
                        top.push(d2, __float_as_int(p.w));
                    }
                }
            }
        }
        const float kth = top.kth(k);
        if (MODE == CLOUD_OUTLIER && kth < r2) break;      // decided: at least k points inside the radius
        const float lb = ((float)rho - 1e-3f) * g.c;       // every unvisited point is at least this far (1e-3: floor() rounding)
        if (lb > 0.0f && kth < lb * lb) break;             // (the host sizes c so that ring max_ring covers the radius)
    }
    if (MODE == CLOUD_KNN) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (j < k) {
                bool in = top.d[j] < r2;
                out.idx[(size_t)qi * k + j] = in ? top.id[j] : -1;
                out.dist[(size_t)qi * k + j] = in ? top.d[j] : __builtin_inff();
            }
        }
    } else if (MODE == CLOUD_OUTLIER) {
        out.mask[qi] = top.kth(k) < r2 ? 1 : 0;           // pcproc.cu:98-105
    } else {
        // pcproc.cu:107-158: neighbours 1..k-1 of the sorted list while inside the radius; mean, covariance, smallest eigenvector,
        // flipped towards the camera.  Fewer than 5 neighbours -> NaN.
        float mx = 0.f, my = 0.f, mz = 0.f, cntf = 0.f;
        bool open = true;
#pragma unroll
        for (int j = 1; j < K; ++j) {
            open = open && (j < k) && (top.d[j] < r2);
            if (open) {
                const float* p = pc + (size_t)top.id[j] * stride;
                mx += p[0]; my += p[1]; mz += p[2];
                cntf += 1.0f;
            }
        }
        float* o = out.normal + (size_t)qi * 3;
        if (cntf < 5.0f) {
            o[0] = o[1] = o[2] = __builtin_nanf("");
            return;
        }
        mx /= cntf; my /= cntf; mz /= cntf;
        float c11 = 0, c12 = 0, c13 = 0, c21 = 0, c22 = 0, c23 = 0, c31 = 0, c32 = 0, c33 = 0;
        open = true;
#pragma unroll
        for (int j = 1; j < K; ++j) {
            open = open && (j < k) && (top.d[j] < r2);
            if (open) {
                const float* p = pc + (size_t)top.id[j] * stride;
                const float px = p[0] - mx, py = p[1] - my, pz = p[2] - mz;
                c11 += px * px; c12 += px * py; c13 += px * pz;
                c21 += py * px; c22 += py * py; c23 += py * pz;
                c31 += pz * px; c32 += pz * py; c33 += pz * pz;
            }
        }
        float nx, ny, nz;
        sym3eig_min(c11, c12, c13, c21, c22, c23, c31, c32, c33, nx, ny, nz);
        const float* self = pc + (size_t)qi * stride;
        const float dt = nx * (self[0] - out.cam[0]) + ny * (self[1] - out.cam[1]) + nz * (self[2] - out.cam[2]);
        if (dt > 0.0f) { nx = -nx; ny = -ny; nz = -nz; }
        o[0] = nx; o[1] = ny; o[2] = nz;
    }
}

// Points that own no cell (NaN / inf / outside the key range): no neighbours.
template <int MODE>
__global__ void __launch_bounds__(DIF_BLOCK) k_cloud_invalid(CloudGrid g, int n, int k, CloudQueryOut out) {
    for (int i = blockIdx.x * DIF_BLOCK + threadIdx.x; i < n; i += gridDim.x * DIF_BLOCK) {
        if (g.slot[i] >= 0) continue;
        if (MODE == CLOUD_KNN) {
            for (int j = 0; j < k; ++j) { out.idx[(size_t)i * k + j] = -1; out.dist[(size_t)i * k + j] = __builtin_inff(); }
        } else if (MODE == CLOUD_OUTLIER) {
            out.mask[i] = 0;
        } else {
            out.normal[(size_t)i * 3 + 0] = out.normal[(size_t)i * 3 + 1] = out.normal[(size_t)i * 3 + 2] = __builtin_nanf("");
        }
    }
}

