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

// Unit eigenvector of the smallest eigenvalue of a symmetric 3x3 matrix M (rows m0, m1, m2), as pcproc.cu:21-96 computes it:
// eigenvalue by the trigonometric closed form on B = (M - q I) / p, eigenvector = the largest of the three pairwise cross products
// of the rows of (M - lambda I).  Operation order follows the reference so that results agree to the last few ulps; its phase shift
// and cosine are evaluated in double precision (M_PI is a double constant there), mirrored here.
struct Row3 { float x, y, z; };

__device__ __forceinline__ Row3 cross3(const Row3& u, const Row3& v) {
    return Row3{u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
}

__device__ __forceinline__ float norm2_3(const Row3& u) { return u.x * u.x + u.y * u.y + u.z * u.z; }

__device__ inline Row3 smallest_eigenvector(Row3 m0, Row3 m1, Row3 m2) {
    const float off2 = m0.y * m0.y + m0.z * m0.z + m1.z * m1.z;
    const float q = (m0.x + m1.y + m2.z) / 3.0f;
    const float dev2 = (m0.x - q) * (m0.x - q) + (m1.y - q) * (m1.y - q) + (m2.z - q) * (m2.z - q) + 2 * off2;
    const float p = sqrtf(dev2 / 6.0f);
    const float ip = 1.0f / p;
    const Row3 b0{ip * (m0.x - q), ip * m0.y, ip * m0.z}, b1{ip * m1.x, ip * (m1.y - q), ip * m1.z}, b2{ip * m2.x, ip * m2.y, ip * (m2.z - q)};
    float half_det = b0.x * b1.y * b2.z + b0.y * b1.z * b2.x + b0.z * b1.x * b2.y - b0.z * b1.y * b2.x - b0.y * b1.x * b2.z - b0.x * b1.z * b2.y;
    half_det = half_det / 2.0f;
    const double third_turn = 2 * 3.14159265358979323846 / 3;
    float phi;
    if (half_det <= -1) phi = (float)(3.14159265358979323846 / 3.0);
    else if (half_det >= 1) phi = 0;
    else phi = acosf(half_det) / 3.0f;
    const float lambda = (float)((double)q + (double)(2 * p) * cos((double)phi + third_turn));
    m0.x -= lambda; m1.y -= lambda; m2.z -= lambda;
    const Row3 c01 = cross3(m0, m1), c02 = cross3(m0, m2), c12 = cross3(m1, m2);
    const float n01 = norm2_3(c01), n02 = norm2_3(c02), n12 = norm2_3(c12);
    // pick order of the reference: c02 replaces c01 when longer; c12 wins whenever it is longer than the better of those two
    const bool take02 = n02 > n01;
    const bool take12 = n12 > (take02 ? n02 : n01);
    const Row3 c = take12 ? c12 : (take02 ? c02 : c01);
    const float len = sqrtf(take12 ? n12 : (take02 ? n02 : n01));
    return Row3{c.x / len, c.y / len, c.z / len};
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
        const Row3 nv = smallest_eigenvector(Row3{c11, c12, c13}, Row3{c21, c22, c23}, Row3{c31, c32, c33});
        float nx = nv.x, ny = nv.y, nz = nv.z;
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

