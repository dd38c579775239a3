/*
 * CPU ORACLE — sparse marching cubes with cross-voxel std-weighted blending.  TEST INFRASTRUCTURE ONLY.
 *
 * Scalar C restatement of the reference CUDA kernel
 *   /root/reference/pytorch/system/ext/marching_cubes/mc_interp_kernel.cu
 *     query_sdf_raw :7-29, get_sdf :34-185, sdf_interp :187-200, meshing_cube :202-320
 * The reference kernel cannot run in the build container (CUDA only), so this file is pinned by known-answer
 * properties only (tests/test_oracle_mc.py): PARITY UNPINNED for this kernel.
 *
 * Differences, by construction:
 *   - output order is canonical (voxel k in valid_blocks order, cell s ascending, table order) instead of the
 *     reference's atomicAdd arrival order (mc_interp_kernel.cu:307);
 *   - every float operation is individually rounded (-ffp-contract=off); nvcc contracts some mul+add pairs of the
 *     reference into FMAs, a <=1-ulp effect on blended values.
 *
 * Build: gcc -O2 -fPIC -shared -std=c99 -ffp-contract=off -o libmc_oracle.so mc_oracle.c -lm
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include "mc_tables_oracle.inc"

/* exported copies so Python can read the tables */
int mc_edge_table[256];
int mc_tri_table[256 * 16];
static void export_tables(void) __attribute__((constructor));
static void export_tables(void) {
    for (int c = 0; c < 256; ++c) {
        mc_edge_table[c] = mc_oracle_edge_table[c];
        for (int i = 0; i < 16; ++i) mc_tri_table[c * 16 + i] = mc_oracle_tri_table[c][i];
    }
}

typedef struct { float x, y; } f2;
typedef struct { float x, y, z, w; } f4;

typedef struct {
    const int64_t* indexer; unsigned nx, ny, nz;
    const int32_t* vbm; unsigned max_vec_num;
    const float* cube_sdf; const float* cube_std; unsigned R;
} ctx_t;

/* mc_interp_kernel.cu:7-29 */
static f2 query_sdf_raw(const ctx_t* c, unsigned bx, unsigned by, unsigned bz, unsigned arx, unsigned ary, unsigned arz) {
    f2 nanv = {NAN, NAN};
    if (bx >= c->nx || by >= c->ny || bz >= c->nz) return nanv;      /* unsigned wrap of -1 lands here (:13) */
    int64_t vec_ind = c->indexer[((size_t)bx * c->ny + by) * c->nz + bz];
    if (vec_ind == -1 || vec_ind >= (int64_t)c->max_vec_num) return nanv;
    int32_t batch_ind = c->vbm[vec_ind];
    if (batch_ind == -1) return nanv;
    size_t R = c->R;
    size_t off = (((size_t)batch_ind * R + arx) * R + ary) * R + arz;
    f2 r = {c->cube_sdf[off], c->cube_std[off]};
    return r;
}

/* mc_interp_kernel.cu:34-185 (STD_W_SDF branch, :32) */
static f2 get_sdf(const ctx_t* c, unsigned r, unsigned bpx, unsigned bpy, unsigned bpz,
                  unsigned rpx, unsigned rpy, unsigned rpz) {
    f2 nanv = {NAN, NAN};
    if (bpx >= c->nx) { bpx = c->nx - 1; rpx = r - 1; }
    if (bpy >= c->ny) { bpy = c->ny - 1; rpy = r - 1; }
    if (bpz >= c->nz) { bpz = c->nz - 1; rpz = r - 1; }
    unsigned rbound = (r - 1) / 2, rstart = r / 2;
    float rmid = r / 2.0f;
    unsigned bp[3] = {bpx, bpy, bpz}, rp[3] = {rpx, rpy, rpz};
    float wm[3], wp[3];
    int bm_[3], rm_[3], bp_[3], rp_[3], zero[3];
    for (int a = 0; a < 3; ++a) {
        if (rp[a] <= rbound) {
            bm_[a] = -1; rm_[a] = (int)r; bp_[a] = 0; rp_[a] = 0;
            wp[a] = (float)rp[a] + rmid; wm[a] = rmid - (float)rp[a];
            zero[a] = 1;
        } else {
            bm_[a] = 0; rm_[a] = 0; bp_[a] = 1; rp_[a] = -(int)r;
            wp[a] = (float)rp[a] - rmid; wm[a] = rmid + (float)r - (float)rp[a];
            zero[a] = 0;
        }
        wm[a] /= (float)r; wp[a] /= (float)r;
        rp[a] += rstart;
    }
    int zero_det = zero[0] * 4 + zero[1] * 2 + zero[2];
    f2 total_weight = {0.0f, 0.0f}, total_sdf = {0.0f, 0.0f};
    for (int k = 0; k < 8; ++k) {                       /* mmm, mmp, mpm, mpp, pmm, pmp, ppm, ppp (:103-181) */
        int sx = (k >> 2) & 1, sy = (k >> 1) & 1, sz = k & 1;
        f2 s = query_sdf_raw(c,
                             bp[0] + (unsigned)(sx ? bp_[0] : bm_[0]), bp[1] + (unsigned)(sy ? bp_[1] : bm_[1]),
                             bp[2] + (unsigned)(sz ? bp_[2] : bm_[2]),
                             rp[0] + (unsigned)(sx ? rp_[0] : rm_[0]), rp[1] + (unsigned)(sy ? rp_[1] : rm_[1]),
                             rp[2] + (unsigned)(sz ? rp_[2] : rm_[2]));
        float w = (sx ? wp[0] : wm[0]) * (sy ? wp[1] : wm[1]) * (sz ? wp[2] : wm[2]);
        if (!isnan(s.x)) {
            total_sdf.x += s.x * w * s.y; total_weight.x += w * s.y;
            total_sdf.y += w * s.y;       total_weight.y += w;
        } else if (zero_det == k) {
            return nanv;
        }
    }
    f2 out = {total_sdf.x / total_weight.x, total_sdf.y / total_weight.y};
    return out;
}

/* Census of sdf_interp's branches (tests/test_oracle_mc.py::test_fma_contraction_bound): [0] edges evaluated, [1..3] early-outs taken
 * (:189, :190, :191), [4..6] edges whose tested quantity lies within 1e-6 of the 1e-5 epsilon of that early-out — the only places where a build
 * with contracted multiply-adds (nvcc's default for the reference) could take another branch than this one. */
static long long census[8];
void mc_oracle_census_read(long long* out, int reset) {
    for (int i = 0; i < 8; ++i) { out[i] = census[i]; if (reset) census[i] = 0; }
}

/* mc_interp_kernel.cu:187-200 */
static f4 sdf_interp(const float p1[3], const float p2[3], float std1, float std2, float v1, float v2) {
    f4 o;
    census[0]++;
    if (fabsf(fabsf(0.0f - v1) - 1.0e-5f) < 1.0e-6f) census[4]++;
    if (fabsf(fabsf(0.0f - v2) - 1.0e-5f) < 1.0e-6f) census[5]++;
    if (fabsf(fabsf(v1 - v2) - 1.0e-5f) < 1.0e-6f) census[6]++;
    if (fabsf(0.0f - v1) < 1.0e-5f) census[1]++;
    else if (fabsf(0.0f - v2) < 1.0e-5f) census[2]++;
    else if (fabsf(v1 - v2) < 1.0e-5f) census[3]++;
    if (fabsf(0.0f - v1) < 1.0e-5f) { o.x = p1[0]; o.y = p1[1]; o.z = p1[2]; o.w = std1; return o; }
    if (fabsf(0.0f - v2) < 1.0e-5f) { o.x = p2[0]; o.y = p2[1]; o.z = p2[2]; o.w = std2; return o; }
    if (fabsf(v1 - v2) < 1.0e-5f)   { o.x = p1[0]; o.y = p1[1]; o.z = p1[2]; o.w = std1; return o; }
    float w2 = (0.0f - v1) / (v2 - v1);
    float w1 = 1 - w2;
    o.x = p1[0] * w1 + p2[0] * w2;
    o.y = p1[1] * w1 + p2[1] * w2;
    o.z = p1[2] * w1 + p2[2] * w2;
    o.w = std1 * w1 + std2 * w2;
    return o;
}

static const int CORNER[8][3] = {{0,0,0},{1,0,0},{1,1,0},{0,1,0},{0,0,1},{1,0,1},{1,1,1},{0,1,1}};   /* :240-270 */
static const int EDGE[12][2] = {{0,1},{1,2},{2,3},{3,0},{4,5},{5,6},{6,7},{7,4},{0,4},{1,5},{2,6},{3,7}}; /* :284-295 */

/* meshing_cube (:202-320) over all (voxel, cell) pairs sequentially.  Returns the number of triangles the
 * kernel would have counted (may exceed `cap`; writes beyond cap are dropped, :308). */
long long mc_oracle_run(const int64_t* indexer, const int64_t* valid_blocks, long long K,
                        const int32_t* vbm, long long V,
                        const float* cube_sdf, const float* cube_std, int R,
                        int nx, int ny, int nz, float max_std, long long cap,
                        float* tri /* (cap,3,3) */, int64_t* tri_id /* (cap) */, float* tri_std /* (cap,3) */) {
    ctx_t c = {indexer, (unsigned)nx, (unsigned)ny, (unsigned)nz, vbm, (unsigned)V, cube_sdf, cube_std, (unsigned)R};
    const unsigned r = (unsigned)R / 2, r3 = r * r * r;
    const float sbs = 1.0f / (float)r;
    long long count = 0;
    for (long long k = 0; k < K; ++k) {
        int64_t vb = valid_blocks[k];
        unsigned bx = (unsigned)((vb / ((int64_t)ny * nz)) % nx), by = (unsigned)((vb / nz) % ny), bz = (unsigned)(vb % nz);
        for (unsigned s = 0; s < r3; ++s) {
            unsigned rx = s / (r * r), ry = (s / r) % r, rz = s % r;
            float pts[8][3]; f2 val[8];
            int dropped = 0;
            for (int q = 0; q < 8 && !dropped; ++q) {
                unsigned cx = rx + CORNER[q][0], cy = ry + CORNER[q][1], cz = rz + CORNER[q][2];
                val[q] = get_sdf(&c, r, bx, by, bz, cx, cy, cz);
                if (isnan(val[q].x)) { dropped = 1; break; }
                pts[q][0] = (float)bx + (float)cx * sbs;
                pts[q][1] = (float)by + (float)cy * sbs;
                pts[q][2] = (float)bz + (float)cz * sbs;
            }
            if (dropped) continue;
            int cube_type = 0;
            for (int q = 0; q < 8; ++q) if (val[q].x < 0) cube_type |= 1 << q;
            int edge_config = mc_oracle_edge_table[cube_type];
            if (edge_config == 0) continue;
            f4 vl[12];
            for (int e = 0; e < 12; ++e)
                if (edge_config & (1 << e)) {
                    int a = EDGE[e][0], b = EDGE[e][1];
                    vl[e] = sdf_interp(pts[a], pts[b], val[a].y, val[b].y, val[a].x, val[b].x);
                }
            for (int i = 0; mc_oracle_tri_table[cube_type][i] != -1; i += 3) {
                f4 vp[3];
                for (int vi = 0; vi < 3; ++vi) vp[vi] = vl[(int)mc_oracle_tri_table[cube_type][i + vi]];
                if (vp[0].w > max_std || vp[1].w > max_std || vp[2].w > max_std) continue;
                long long t = count++;
                if (t < cap) {
                    for (int vi = 0; vi < 3; ++vi) {
                        tri[(t * 3 + vi) * 3 + 0] = vp[vi].x;
                        tri[(t * 3 + vi) * 3 + 1] = vp[vi].y;
                        tri[(t * 3 + vi) * 3 + 2] = vp[vi].z;
                        tri_std[t * 3 + vi] = vp[vi].w;
                    }
                    tri_id[t] = vb;
                }
            }
        }
    }
    return count;
}
