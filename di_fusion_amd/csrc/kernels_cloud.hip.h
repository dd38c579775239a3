// Point-cloud neighbourhood ops of the tracker's pre-processing: radius outlier removal and PCA normals
// over the k nearest neighbours.  Reference: ext/pcproc/pcproc.cu:98-209 on top of a FLANN-derived CUDA kd-tree
// (cuda_kdtree.cu:644-1260).  Here: a uniform grid of cell size c hashed into an open-addressing table (no bounding box, no
// sort) and an exact ring-by-ring search that stops as soon as the k-th distance is below the lower bound of everything not
// yet visited, or that bound passes the search radius.  Results are the exact kNN under the total order (d2, index), so
// they do not depend on the (racy) order in which points land inside a cell.
#pragma once

#define CLOUD_EMPTY 0xFFFFFFFFFFFFFFFFull
#define CLOUD_OFF (1 << 20)

// One table entry = one 16-byte load: the cell key, its point count MINUS ONE (the table is cleared to all-ones, the first point of a
// cell brings the count to 0) and, after placement, one past the cell's last row in `sorted`.
struct CloudEntry { unsigned long long key; int cnt_m1; int end; };

struct CloudGrid {
    CloudEntry* tab;            // [T]
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

// First-probe slot of a cell.  The cells of one 8 x 8 x 8 block share a 512-slot window of the table (which window: a multiplicative hash of
// the block's coordinates) and sit in it in x-y-z order, cyclically shifted by a second hash of the block — so that two blocks that land in the
// same window (planes of one wall have the same local pattern) do not lie on top of each other.  Rows of `sorted` are handed out in SLOT order
// (CloudStartFunctor), so this makes rows — and with them the query threads, one per row — spatially coherent: consecutive workgroups walk the
// same neighbour cells and find their rows in L2 instead of in HBM (a plain hash of the cell scattered neighbouring cells over the whole
// table: 257 MB of counter traffic per 307 k-point normal estimation for a 15 MB working set).  The table has >= 4,096 slots (carve_cloud).
__device__ __forceinline__ unsigned cloud_hash(const CloudGrid& g, unsigned long long key) {
    const unsigned cz = (unsigned)key & 0x1FFFFFu, cy = (unsigned)(key >> 21) & 0x1FFFFFu, cx = (unsigned)(key >> 42) & 0x3FFFFFu;
    const unsigned long long blk = ((unsigned long long)(cx >> 3) << 36) | ((unsigned long long)(cy >> 3) << 18) | (unsigned long long)(cz >> 3);
    const unsigned long long hb = blk * 0x9E3779B97F4A7C15ull;
    const unsigned window = (unsigned)(hb >> (g.shift + 9));                 // log2(T) - 9 bits
    const unsigned local = ((cx & 7u) << 6) | ((cy & 7u) << 3) | (cz & 7u);
    const unsigned rot = (unsigned)(hb >> 11) & 511u;
    return (window << 9) | ((local + rot) & 511u);
}
// Where the probe sequence goes when the home slot is taken by another cell (two blocks in one window): OUT of the window, to a plain hash of
// the cell and linearly on from there.  Continuing inside the window would run through the block's own occupied slots — and most look-ups of a
// query are for EMPTY neighbour cells, which must reach a free slot to know that they are empty.
__device__ __forceinline__ unsigned cloud_hash2(const CloudGrid& g, unsigned long long key) {
    return (unsigned)((key * 0x9E3779B97F4A7C15ull) >> g.shift);
}

__global__ void __launch_bounds__(DIF_BLOCK) k_cloud_insert(CloudGrid g, const float* __restrict__ pc, int n, int stride) {
    for (int i = blockIdx.x * DIF_BLOCK + threadIdx.x; i < n; i += gridDim.x * DIF_BLOCK) {
        const float* p = pc + (size_t)i * stride;
        int cx, cy, cz;
        int slot = -1;
        if (cloud_cell(g, p[0], p[1], p[2], cx, cy, cz)) {
            unsigned long long key = cloud_key(cx, cy, cz);
            unsigned h = cloud_hash(g, key);
            bool home = true;
            while (true) {
                unsigned long long prev = atomicCAS(&g.tab[h].key, CLOUD_EMPTY, key);
                if (prev == CLOUD_EMPTY || prev == key) break;
                h = home ? cloud_hash2(g, key) : ((h + 1) & g.mask);
                home = false;
            }
            slot = (int)h;
            atomicAdd(&g.tab[h].cnt_m1, 1);
        }
        g.slot[i] = slot;
    }
}

struct CloudStartFunctor {
    CloudEntry* tab;
    __device__ int count(int i) const { return tab[i].key == CLOUD_EMPTY ? 0 : tab[i].cnt_m1 + 1; }
    __device__ void emit(int i, int off) const { tab[i].end = off; }     // becomes the placement cursor
    __device__ void finish(int) const {}
};

enum { CLOUD_KNN = 0, CLOUD_OUTLIER = 1, CLOUD_NORMAL = 2 };

struct CloudQueryOut {
    int* idx;            // KNN: (n, k)
    float* dist;         // KNN: (n, k)
    uint8_t* mask;       // OUTLIER: (n)
    float* normal;       // NORMAL: (n, 3)
    float cam[3];
};

// Rows grouped by cell; a point that owns no cell (NaN / inf / outside the key range) has no neighbours and gets its answer here.
template <int MODE>
__global__ void __launch_bounds__(DIF_BLOCK) k_cloud_place(CloudGrid g, const float* __restrict__ pc, int n, int stride, int k, CloudQueryOut out) {
    for (int i = blockIdx.x * DIF_BLOCK + threadIdx.x; i < n; i += gridDim.x * DIF_BLOCK) {
        int slot = g.slot[i];
        if (slot < 0) {
            if (MODE == CLOUD_KNN) {
                for (int j = 0; j < k; ++j) { out.idx[(size_t)i * k + j] = -1; out.dist[(size_t)i * k + j] = __builtin_inff(); }
            } else if (MODE == CLOUD_OUTLIER) {
                out.mask[i] = 0;
            } else {
                out.normal[(size_t)i * 3 + 0] = out.normal[(size_t)i * 3 + 1] = out.normal[(size_t)i * 3 + 2] = __builtin_nanf("");
            }
            continue;
        }
        const float* p = pc + (size_t)i * stride;
        int pos = atomicAdd(&g.tab[slot].end, 1);
        g.sorted[pos] = make_float4(p[0], p[1], p[2], __int_as_float(i));
    }
}

// ---- bounded top-K under the total order (d2, index) -------------------------------------------------------------------
// Entries are 64-bit keys (float bits of d2 in the high word — d2 >= 0 and never NaN, so the bit patterns order like the values — and
// the point index in the low word): one unsigned compare decides the order.
__device__ __forceinline__ unsigned long long cloud_pack(float d2, int idx) { return ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)idx; }

template <int K>
struct TopK {
    // The list has K slots (K = 8, 16, 32: the compiled sizes) for a run-time k <= K.  The K - k LEADING slots hold a sentinel that
    // nothing sorts below (key 0), so the k best real entries always sit in slots K - k .. K - 1 and the k-th best is simply the LAST slot:
    // no run-time index into the list anywhere in the search loop.  (Selecting slot k - 1 of a plain list with a run-time k made the compiler
    // park the whole list in scratch memory and load the one entry back — K x 8 bytes stored per batch of cells.)  Fully unrolled accesses
    // only: the list stays in registers.
    unsigned long long kv[K];
    __device__ __forceinline__ void init(int k) {
#pragma unroll
        for (int j = 0; j < K; ++j) kv[j] = (j < K - k) ? 0ull : cloud_pack(__builtin_inff(), 0x7FFFFFFF);
    }
    __device__ __forceinline__ unsigned long long worst() const { return kv[K - 1]; }
    __device__ __forceinline__ void insert(unsigned long long ck) {        // ck < worst() is the caller's business; branch-free
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const bool lt = ck < kv[j];                                     // (never true for a sentinel slot)
            const unsigned long long t = kv[j];
            kv[j] = lt ? ck : t;
            ck = lt ? t : ck;
        }
    }
    // slot j of the K (compile-time index); its rank among the real entries is j - (K - k)
    __device__ __forceinline__ float d(int j) const { return __uint_as_float((unsigned)(kv[j] >> 32)); }
    __device__ __forceinline__ int id(int j) const { return (int)(unsigned)kv[j]; }
    __device__ __forceinline__ float kth() const { return __uint_as_float((unsigned)(kv[K - 1] >> 32)); }      // distance of the k-th best so far
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

// One thread per point, in cell order (so a wave's lanes walk the same few cells).  Ring rho = the shell of cells at Chebyshev
// distance rho from the query's cell; once rings 0..rho are done every unvisited point is farther than rho*c.
// The search is a chain of dependent look-ups (cell -> table entry -> rows) over 27-125 cells per point, and a frame's cloud is barely
// one wave per SIMD: what a thread has in flight decides the run time.  So a shell is walked CLOUD_NB cells at a time — their first-probe
// table entries (one 16-byte load each: key, count, row range) are requested together — and the rows of the batch's non-empty cells are
// walked as one flat sequence, CLOUD_U row loads at a time (the cells' row ranges wait in a per-thread strip of LDS).
#define CLOUD_NB 16
#define CLOUD_U 4

template <int K, int MODE>
__global__ void __launch_bounds__(DIF_BLOCK) k_cloud_query(CloudGrid g, const float* __restrict__ pc, int n, int stride, int k, float radius,
                                                           int max_ring, CloudQueryOut out) {
    __shared__ int2 s_run[CLOUD_NB * DIF_BLOCK];           // [cell of the batch][thread]: (first row - rows before it in the batch, rows before it)
    const int row = blockIdx.x * DIF_BLOCK + threadIdx.x;
    // rows of `sorted` beyond the finite points do not exist; the finite count is the table's total
    if (row >= n) return;
    // non-finite points never got a row: k_cloud_place answered for them
    const float4 q = g.sorted[row];
    const int qi = __float_as_int(q.w);
    if (qi < 0) return;                                    // unused tail row (sorted is pre-filled with index -1)
    int cx, cy, cz;
    cloud_cell(g, q.x, q.y, q.z, cx, cy, cz);
    const float r2 = radius * radius;
    const uint4* __restrict__ tab4 = reinterpret_cast<const uint4*>(g.tab);
    int2* const run = s_run + threadIdx.x;
    TopK<K> top;
    top.init(k);
    const int lead = K - k;                                 // sentinel slots in front of the list
    int inside = 0;
    for (int rho = 0; rho <= max_ring; ++rho) {
        int dx = -rho, dy = -rho, dz = -rho;               // the shell in the order of three nested loops, interior columns reduced to their end caps
        bool more = true;
        while (more) {
            uint4 ent[CLOUD_NB];
            int off[CLOUD_NB];
            // kNN / normals: a cell whose nearest corner is strictly farther than the k-th distance so far (ties go by index) is not looked up
            // at all.  The cell's box from the integer coordinates, shrunk by the same 1e-3 c that the ring bound below allows for floor()
            // rounding.  (For the outlier count the same test against the radius costs more than it saves: measured.)
            const float far2 = top.kth();
#pragma unroll
            for (int b = 0; b < CLOUD_NB; ++b) {
                off[b] = -1;
                ent[b] = uint4{0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u};
                if (more) {
                    bool look = true;
                    if (MODE != CLOUD_OUTLIER) {
                        const float gx = dx < 0 ? q.x - (float)(cx + dx + 1) * g.c : dx > 0 ? (float)(cx + dx) * g.c - q.x : 0.0f;
                        const float gy = dy < 0 ? q.y - (float)(cy + dy + 1) * g.c : dy > 0 ? (float)(cy + dy) * g.c - q.y : 0.0f;
                        const float gz = dz < 0 ? q.z - (float)(cz + dz + 1) * g.c : dz > 0 ? (float)(cz + dz) * g.c - q.z : 0.0f;
                        const float slack = 1e-3f * g.c;
                        const float hx = fmaxf(gx - slack, 0.0f), hy = fmaxf(gy - slack, 0.0f), hz = fmaxf(gz - slack, 0.0f);
                        look = (hx * hx + hy * hy) + hz * hz <= far2;
                    }
                    if (look) {
                        off[b] = ((dx + 64) << 16) | ((dy + 64) << 8) | (dz + 64);
                        ent[b] = tab4[cloud_hash(g, cloud_key(cx + dx, cy + dy, cz + dz))];
                    }
                    const bool face = (dx == -rho) || (dx == rho) || (dy == -rho) || (dy == rho);
                    dz += (face || rho == 0) ? 1 : 2 * rho;
                    if (dz > rho) { dz = -rho; if (++dy > rho) { dy = -rho; if (++dx > rho) more = false; } }
                }
            }
            // ---- which of them exist, and where their rows are ----
            int m = 0, tot = 0;
#pragma unroll
            for (int b = 0; b < CLOUD_NB; ++b) {
                if (off[b] < 0) continue;
                const unsigned long long key = cloud_key(cx + (off[b] >> 16) - 64, cy + ((off[b] >> 8) & 0xFF) - 64, cz + (off[b] & 0xFF) - 64);
                unsigned long long got = ((unsigned long long)ent[b].y << 32) | ent[b].x;
                int c = (int)ent[b].z + 1, e = (int)ent[b].w;
                if (got != key && got != CLOUD_EMPTY) {                 // the first probe hit another cell's entry (rare): walk on
                    unsigned h = cloud_hash2(g, key);
                    while (true) {
                        const uint4 v = tab4[h];
                        got = ((unsigned long long)v.y << 32) | v.x;
                        c = (int)v.z + 1; e = (int)v.w;
                        if (got == key || got == CLOUD_EMPTY) break;
                        h = (h + 1) & g.mask;
                    }
                }
                if (got == key) {
                    run[m * DIF_BLOCK] = make_int2(e - c - tot, tot);
                    tot += c;
                    ++m;
                }
            }
            // ---- the batch's rows as one sequence ----
            int cur = 0, base = 0, lim = 0;
            if (m > 0) { base = run[0].x; lim = m > 1 ? run[DIF_BLOCK].y : tot; }
            for (int t0 = 0; t0 < tot; t0 += CLOUD_U) {
                float4 p[CLOUD_U];
#pragma unroll
                for (int u = 0; u < CLOUD_U; ++u) {
                    const int t = t0 + u;
                    if (t < tot) {
                        while (t >= lim) {
                            ++cur;
                            base = run[cur * DIF_BLOCK].x;
                            lim = cur + 1 < m ? run[(cur + 1) * DIF_BLOCK].y : tot;
                        }
                        p[u] = g.sorted[base + t];
                    }
                }
#pragma unroll
                for (int u = 0; u < CLOUD_U; ++u) {
                    if (t0 + u < tot) {
                        const float ex = p[u].x - q.x, ey = p[u].y - q.y, ez = p[u].z - q.z;
                        const float d2 = (ex * ex + ey * ey) + ez * ez;       // CudaL2::dist (cuda_kdtree.cu:1152-1155)
                        if (MODE == CLOUD_OUTLIER) inside += d2 < r2 ? 1 : 0;  // "the k-th distance is inside the radius" = "k points are": no list needed
                        else {
                            const unsigned long long ck = cloud_pack(d2, __float_as_int(p[u].w));
                            if (ck < top.worst()) top.insert(ck);
                        }
                    }
                }
            }
            if (MODE == CLOUD_OUTLIER && inside >= k) break;
        }
        if (MODE == CLOUD_OUTLIER) {
            if (inside >= k) break;                        // decided: at least k points inside the radius
            continue;                                      // (ring max_ring covers the radius: the count is complete when the loop ends)
        }
        const float kth = top.kth();
        const float lb = ((float)rho - 1e-3f) * g.c;       // every unvisited point is at least this far (1e-3: floor() rounding)
        if (lb > 0.0f && kth < lb * lb) break;             // (the host sizes c so that ring max_ring covers the radius)
    }
    if (MODE == CLOUD_KNN) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (j >= lead) {
                bool in = top.d(j) < r2;
                out.idx[(size_t)qi * k + (j - lead)] = in ? top.id(j) : -1;
                out.dist[(size_t)qi * k + (j - lead)] = in ? top.d(j) : __builtin_inff();
            }
        }
    } else if (MODE == CLOUD_OUTLIER) {
        out.mask[qi] = inside >= k ? 1 : 0;               // pcproc.cu:98-105 (k-th distance < radius^2)
    } else {
        // pcproc.cu:107-158: neighbours 1..k-1 of the sorted list while inside the radius; mean, covariance, smallest eigenvector,
        // flipped towards the camera.  Fewer than 5 neighbours -> NaN.
        float mx = 0.f, my = 0.f, mz = 0.f, cntf = 0.f;
        bool open = true;
#pragma unroll
        for (int j = 1; j < K; ++j) {                      // real entries 1 .. k-1 (entry 0 is the query itself) sit in slots lead+1 .. K-1
            if (j <= lead) continue;
            open = open && (top.d(j) < r2);
            if (open) {
                const float* p = pc + (size_t)top.id(j) * stride;
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
            if (j <= lead) continue;
            open = open && (top.d(j) < r2);
            if (open) {
                const float* p = pc + (size_t)top.id(j) * stride;
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
