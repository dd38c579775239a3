// a15 sparse marching cubes, a16 mesh-cache log, a17 point query, multi-GPU record export / merge  (part of libdifusion; included by difusion.hip inside its anonymous namespace)
#pragma once

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
    float* corner_cache; int64_t corner_stride;   // optional [K][corner_stride >= 2(r+1)^3]: the count pass parks the blended corners here, the emit pass picks them up
    int* log_counters;              // mesh-cache path: the count pass freezes DIF_C_CACHE_KEPT = DIF_C_CACHE_T (log length before this call)
    int64_t new_limit;              // triangles this call may emit (max_n_triangles)
    int scale; float vs, bx, by, bz;
    // Fused scan (mesh-cache path): the count pass adds every voxel's triangle count to chunk_sum[k >> 8] and super_sum[k >> 16]
    // (both idle 0), the emit pass derives a voxel's output offset from < 256 super sums + < 256 chunk sums + < 256 counts and does
    // the log bookkeeping itself — no scan launch between the two passes.
    int32_t* chunk_sum; int32_t* super_sum; int32_t* tri_start; int32_t* tri_n;
    int* grid_tot;                  // extract path: the grid bitmap's scan totals, returned to idle 0 here (the voxel scan has consumed them)
    // one-pass kernel only: the caller's copy of this call's new triangles (dif_extract_buffers_t.out_*; pinned host memory allowed),
    // written by the wave that emits them, so the transfer runs while the rest of the launch is still counting and emitting
    float* out_tri; int64_t* out_id; float* out_std; int64_t out_capacity;
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
    const int R = 2 * r;                    // (== a.R; a compile-time constant where the caller's r is one)
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
    // All eight (sdf, std) samples are requested before any of them is looked at: sixteen independent loads in flight.  (With the loads inside
    // the blend loop, each behind the previous sample's "is it NaN / is it the voxel's own" test, they went out one after the other — eight
    // dependent round trips per corner, 6.4 us of a stream frame's 21 us launch: profiles/r04_experiments.md.)  The blend itself is unchanged:
    // same terms, same order; a missing own sample still drops the corner, whatever the other seven are.
    float sv[8], dv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int sx = (k >> 2) & 1, sy = (k >> 1) & 1, sz = k & 1;
        const int ddx = sx ? dp[0] : dm[0], ddy = sy ? dp[1] : dm[1], ddz = sz ? dp[2] : dm[2];
        const int b = nb[(ddx + 1) * 9 + (ddy + 1) * 3 + (ddz + 1)];
        sv[k] = __builtin_nanf(""); dv[k] = 0.0f;
        if (b >= 0) {                            // (predicated on the neighbour table only: nothing here waits for a loaded value)
            const int64_t off = (((int64_t)b * R + (sx ? ip[0] : im[0])) * R + (sy ? ip[1] : im[1])) * R + (sz ? ip[2] : im[2]);
            sv[k] = a.cube_sdf[off];
            dv[k] = a.cube_std[off];
        }
    }
    float ts = 0.0f, tw = 0.0f, tsd = 0.0f, twd = 0.0f;     // total_sdf.x, total_weight.x, total_sdf.y, total_weight.y
    bool own_missing = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int sx = (k >> 2) & 1, sy = (k >> 1) & 1, sz = k & 1;
        const float s = sv[k], d = dv[k];
        const float w = (sx ? wp[0] : wm[0]) * (sy ? wp[1] : wm[1]) * (sz ? wp[2] : wm[2]);
        if (s == s) {
            ts += s * w * d; tw += w * d;
            tsd += w * d;    twd += w;
        } else if (zero_det == k) {
            own_missing = true;
        }
    }
    if (own_missing) return false;
    sdf = ts / tw;
    sd = tsd / twd;
    return sdf == sdf;
}

struct V4 { float x, y, z, w; };

// An edge vertex as the cell keeps it in LDS: the weight of the edge's second end point, 4 bytes (16 as (x, y, z, std) until round 4 — the
// LDS per wave is what bounds the occupancy of these kernels).  sdf_interp (mc_interp_kernel.cu:187-200) returns p1 * w1 + p2 * w2 and
// s1 * w1 + s2 * w2 with w1 = 1 - w2, or one end point unchanged in its three early-outs: those are w2 = 0 and w2 = 1 of the same
// expressions, exactly (p * 1 + q * 0 = p for the finite, non-negative lattice coordinates and stds), so position and std are rebuilt
// from w2 where a triangle is written (mc_vertex) and come out bit-identical to interpolating them on the spot.  What the count needs of
// the std — is it above max_std? — is one bit per edge, kept in a register (mc_eval_cell's `ok`).
struct V2 { float w2, sd; };

__device__ __forceinline__ V2 mc_interp(float s1, float s2, float v1, float v2) {
    if (fabsf(0.0f - v1) < 1.0e-5f) return V2{0.0f, s1};
    if (fabsf(0.0f - v2) < 1.0e-5f) return V2{1.0f, s2};
    if (fabsf(v1 - v2) < 1.0e-5f) return V2{0.0f, s1};
    float w2 = (0.0f - v1) / (v2 - v1);
    float w1 = 1 - w2;
    return V2{w2, s1 * w1 + s2 * w2};
}

// The six lattice coordinates a cell's corners are made of, in voxel units (the corner positions of mc_interp_kernel.cu:236-262: the low and
// the high one per axis), the cell's first corner in the (r+1)^3 corner arrays, and the vertex on edge e rebuilt from its weight.  Which
// corner an end point is comes from a table of three bits (dx, dy, dz) per edge and end: the emit loop runs with one wave per SIMD in a
// stream, where every instruction is latency.
struct CellBox { float x[2], y[2], z[2]; int c0, r1; };
__device__ __forceinline__ CellBox mc_cell_box(int bx, int by, int bz, int rx, int ry, int rz, int r) {
    const float sbs = 1.0f / (float)r;
    return CellBox{{(float)bx + (float)rx * sbs, (float)bx + (float)(rx + 1) * sbs}, {(float)by + (float)ry * sbs, (float)by + (float)(ry + 1) * sbs},
                   {(float)bz + (float)rz * sbs, (float)bz + (float)(rz + 1) * sbs}, (rx * (r + 1) + ry) * (r + 1) + rz, r + 1};
}
constexpr unsigned long long mc_end_bits(bool second) {
    const int ea[12] = {0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3}, eb[12] = {1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7};
    unsigned long long t = 0;
    for (int e = 0; e < 12; ++e) {
        const int q = second ? eb[e] : ea[e];
        const unsigned long long dx = (q == 1 || q == 2 || q == 5 || q == 6), dy = (q == 2 || q == 3 || q == 6 || q == 7), dz = (q >= 4);
        t |= (dx | (dy << 1) | (dz << 2)) << (3 * e);
    }
    return t;
}
// (vsd: the edge stds kept beside the weights — the one-group-per-workgroup mode has the LDS for them — or NULL: rebuilt from the corners)
__device__ __forceinline__ V4 mc_vertex(float w2, int e, const CellBox& c, const float* __restrict__ c_std, const float* __restrict__ vsd = nullptr) {
    constexpr unsigned long long TA = mc_end_bits(false), TB = mc_end_bits(true);
    const unsigned sa = (unsigned)(TA >> (3 * e)), sb = (unsigned)(TB >> (3 * e));
    const float w1 = 1 - w2;
    const float ax = (sa & 1) ? c.x[1] : c.x[0], ay = (sa & 2) ? c.y[1] : c.y[0], az = (sa & 4) ? c.z[1] : c.z[0];
    const float cx = (sb & 1) ? c.x[1] : c.x[0], cy = (sb & 2) ? c.y[1] : c.y[0], cz = (sb & 4) ? c.z[1] : c.z[0];
    if (vsd) return V4{ax * w1 + cx * w2, ay * w1 + cy * w2, az * w1 + cz * w2, vsd[e * 64]};
    const float s1 = c_std[c.c0 + ((sa & 1) ? c.r1 * c.r1 : 0) + ((sa & 2) ? c.r1 : 0) + ((sa >> 2) & 1)];
    const float s2 = c_std[c.c0 + ((sb & 1) ? c.r1 * c.r1 : 0) + ((sb & 2) ? c.r1 : 0) + ((sb >> 2) & 1)];
    return V4{ax * w1 + cx * w2, ay * w1 + cy * w2, az * w1 + cz * w2, s1 * w1 + s2 * w2};
}
// does triangle (e0, e1, e2) pass max_std (mc_interp_kernel.cu:304: dropped when any of its three stds is above)?  ok = one bit per edge
__device__ __forceinline__ bool mc_tri_ok(unsigned ok, unsigned long long t3) {
    return (((ok >> (unsigned)(t3 & 0xF)) & (ok >> (unsigned)((t3 >> 4) & 0xF)) & (ok >> (unsigned)((t3 >> 8) & 0xF))) & 1u) != 0u;
}

// One cell of a voxel (lane = cell): reads its 8 blended corners from LDS, writes the weights of the cell's edge vertices to vl[edge * 64],
// returns the number of triangles that survive max_std (mc_interp_kernel.cu:202-320); tri_row = the packed triangle-table row (~0 if none),
// ok = per edge: its interpolated std is not above max_std.
__device__ __forceinline__ int mc_eval_cell(const McArgs& a, const float* __restrict__ c_sdf, const float* __restrict__ c_std, float* __restrict__ vl,
                                            int r, int cell, unsigned long long& tri_row, unsigned& ok, float* __restrict__ vsd = nullptr) {
    const int r1 = r + 1;
    const int rx = cell / (r * r), ry = (cell / r) % r, rz = cell % r;
    float val[8], sdv[8];
    bool dropped = false;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int dx = (q == 1 || q == 2 || q == 5 || q == 6), dy = (q == 2 || q == 3 || q == 6 || q == 7), dz = (q >= 4);
        const int ci = ((rx + dx) * r1 + (ry + dy)) * r1 + (rz + dz);
        val[q] = c_sdf[ci]; sdv[q] = c_std[ci];
        dropped |= !(val[q] == val[q]);
    }
    tri_row = ~0ull;
    ok = 0u;
    if (dropped) return 0;
    int cube_type = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) cube_type |= (val[q] < 0.0f) ? (1 << q) : 0;
    const int edge_config = c_mc_edge_table[cube_type];
    if (!edge_config) return 0;
    const int ea[12] = {0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3}, eb[12] = {1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7};
#pragma unroll
    for (int e = 0; e < 12; ++e)
        if (edge_config & (1 << e)) {
            const V2 v = mc_interp(sdv[ea[e]], sdv[eb[e]], val[ea[e]], val[eb[e]]);
            vl[e * 64] = v.w2;
            if (vsd) vsd[e * 64] = v.sd;
            ok |= (v.sd > a.max_std) ? 0u : (1u << e);
        }
    tri_row = c_mc_tri_packed[cube_type];
    int ntri = 0;
    for (unsigned long long t3 = tri_row; (t3 & 0xF) != 0xF; t3 >>= 12) ntri += mc_tri_ok(ok, t3) ? 1 : 0;     // :304
    return ntri;
}

// One wave per dirty voxel.  Phase 1: the (r+1)^3 blended corner values are computed ONCE into LDS (the reference
// recomputes each corner for up to 8 cells).  Phase 2: lane = cell; EMIT=false counts the triangles that survive
// max_std, EMIT=true writes them at tri_offset[k] + wave-prefix (canonical order: voxel, cell, table order).
#define MC_WAVE_LDS_FLOATS(nc) (((2 * (nc) + 32 + 3) & ~3) + 12 * 64)

template <bool EMIT>
__global__ void __launch_bounds__(DIF_BLOCK) k_marching_cubes(McArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int r = a.R / 2, r1 = r + 1, nc = r1 * r1 * r1, r3 = r * r * r;
    const int lane = lane_id(), wid = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    float* c_sdf = lds + (size_t)wid * MC_WAVE_LDS_FLOATS(nc);
    float* c_std = c_sdf + nc;
    int* nb = reinterpret_cast<int*>(c_std + nc);            // 27 (+pad to 32)
    // edge vertices of the lane's cell, [edge][lane] weights: indexed by the triangle table at run time, so they live in
    // LDS — as a per-lane array they were spilled to scratch memory (13 KB of scratch traffic per voxel).  Measured alternatives
    // on 2.1 M voxels (count + emit ms): scratch array 9.4 + 18.2, (x,y,z,std) in LDS 11.3 + 14.2, vertices recomputed per triangle
    // 11.9 + 19.2, LDS-staged neighbour samples instead of gathers 14.2 + 16.7.  PMC (profiles/r01_pmc_sq_stress.json): the waves sit
    // parked on memory 54-67 % of their cycles at ~3 waves per SIMD — latency-bound, and the LDS per wave is what caps the occupancy:
    // 3 KB per wave as weights since round 4 (mc_interp / mc_vertex), 12 KB before.
    float* vl = lds + (size_t)wid * MC_WAVE_LDS_FLOATS(nc) + ((2 * nc + 32 + 3) & ~3) + lane;
    const int64_t K = a.K_ptr ? (int64_t)(*a.K_ptr) : a.K_static;
    if (!EMIT && a.log_counters && blockIdx.x == 0 && threadIdx.x == 0) a.log_counters[DIF_C_CACHE_KEPT] = a.log_counters[DIF_C_CACHE_T];
    if (!EMIT && a.grid_tot && blockIdx.x == 0)
        for (int t = (int)threadIdx.x; t < 1024; t += (int)blockDim.x) a.grid_tot[t] = 0;
    if (EMIT && a.chunk_sum && blockIdx.x == 0 && wid == 0) {        // triangles of this call = sum of all chunk sums (map.py:695 counts them on the host)
        int tot = 0;
        for (int c = lane; c < (int)((K + 65535) >> 16); c += 64) tot += a.super_sum[c];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) tot += __shfl_xor(tot, d);
        if (lane == 0) a.log_counters[DIF_C_T] = tot;
    }
    for (int64_t k = (int64_t)blockIdx.x * wpb + wid; k < K; k += (int64_t)gridDim.x * wpb) {
        if (EMIT && a.tri_count[k] == 0) continue;          // nothing to write for this voxel: the count pass has already said so
        const int64_t vb = a.valid_blocks[k];
        int voxel_offset = 0;                                // index of the voxel's first triangle among this call's triangles
        if (EMIT) {
            if (a.chunk_sum) {
                const int s0 = (int)(k >> 16), c0 = (int)(k >> 8);
                int part = 0;
                for (int c = lane; c < s0; c += 64) part += a.super_sum[c];
                for (int c = (s0 << 8) + lane; c < c0; c += 64) part += a.chunk_sum[c];
                for (int j = (c0 << 8) + lane; j < (int)k; j += 64) part += a.tri_count[j];
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
                voxel_offset = part;
                // mesh-cache log: this voxel's previous batch dies, the voxel points at its new one (map.py:708-709)
                const int64_t slot = a.indexer[vb], log_n = a.log_counters[DIF_C_CACHE_KEPT];
                const int old_n = a.tri_n[slot], old_s = a.tri_start[slot];
                for (int j = lane; j < old_n; j += 64) a.tri_alive[old_s + j] = 0;
                int64_t n_new = a.tri_count[k];
                if (voxel_offset + n_new > a.new_limit) n_new = a.new_limit > voxel_offset ? a.new_limit - voxel_offset : 0;      // truncated by max_n_triangles
                if (log_n + voxel_offset + n_new > a.max_triangles) n_new = a.max_triangles > log_n + voxel_offset ? a.max_triangles - (log_n + voxel_offset) : 0;
                __builtin_amdgcn_wave_barrier();             // every lane has read tri_n / tri_start
                if (lane == 0) {
                    a.tri_start[slot] = (int)(log_n + voxel_offset);
                    a.tri_n[slot] = (int)n_new;
                    if (old_n) atomicAdd(a.log_counters + DIF_C_CACHE_DEAD, old_n);
                }
            } else {
                voxel_offset = a.tri_offset[k];
            }
        }
        const int bx = (int)((vb / ((int64_t)a.ny * a.nz)) % a.nx), by = (int)((vb / a.nz) % a.ny), bz = (int)(vb % a.nz);
        bool any_neg = true, any_pos = true;
        if (EMIT && a.corner_cache) {                        // blended corners as the count pass left them (c_sdf and c_std are contiguous)
            const float* cc = a.corner_cache + k * a.corner_stride;
            for (int c = lane; c < 2 * nc; c += 64) c_sdf[c] = cc[c];
        } else {
            if (lane < 27) nb[lane] = mc_batch_of(a, bx + lane / 9 - 1, by + (lane / 3) % 3 - 1, bz + lane % 3 - 1);
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0): nb[] visible to the whole wave
            any_neg = false; any_pos = false;
            for (int c = lane; c < nc; c += 64) {
                float s, d;
                bool ok = mc_corner(a, nb, r, c / (r1 * r1), (c / r1) % r1, c % r1, s, d);
                c_sdf[c] = ok ? s : __builtin_nanf("");
                c_std[c] = ok ? d : 0.0f;
                any_neg |= ok && s < 0.0f;
                any_pos |= ok && !(s < 0.0f);
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        int voxel_total = 0;
        // a cell needs corners of both signs to produce a triangle: most dirty voxels off the surface stop here
        const bool crossing = __ballot(any_neg) != 0ull && __ballot(any_pos) != 0ull;
        if (!EMIT && a.corner_cache && crossing) {
            float* cc = a.corner_cache + k * a.corner_stride;
            for (int c = lane; c < 2 * nc; c += 64) cc[c] = c_sdf[c];
        }
        for (int s0 = 0; crossing && s0 < r3; s0 += 64) {
            const int s = s0 + lane;
            unsigned long long tri_row = ~0ull;
            unsigned ok = 0u;
            const int ntri = (s < r3) ? mc_eval_cell(a, c_sdf, c_std, vl, r, s, tri_row, ok) : 0;
            const int incl = wave_incl_scan(ntri);
            const int chunk_total = __shfl(incl, 63);
            if (EMIT && ntri > 0) {
                int64_t tl = (int64_t)voxel_offset + voxel_total + (incl - ntri);         // index among this call's triangles
                int64_t t = tl + (a.base_ptr ? (int64_t)(*a.base_ptr) : 0);
                const CellBox box = mc_cell_box(bx, by, bz, s / (r * r), (s / r) % r, s % r, r);
                for (unsigned long long t3 = tri_row; (t3 & 0xF) != 0xF; t3 >>= 12) {
                    const int e0 = (int)(t3 & 0xF), e1 = (int)((t3 >> 4) & 0xF), e2 = (int)((t3 >> 8) & 0xF);
                    if (!mc_tri_ok(ok, t3)) continue;
                    if (tl < a.new_limit && t < a.max_triangles) {
                        V4 vv[3] = {mc_vertex(vl[e0 * 64], e0, box, c_std), mc_vertex(vl[e1 * 64], e1, box, c_std), mc_vertex(vl[e2 * 64], e2, box, c_std)};
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
        if (!EMIT && lane == 0) {
            a.tri_count[k] = voxel_total;
            if (a.chunk_sum && voxel_total) { atomicAdd(a.chunk_sum + (k >> 8), voxel_total); atomicAdd(a.super_sum + (k >> 16), voxel_total); }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- one-pass variant for the stream (mesh-cache path, r^3 <= 64 cells per voxel) ---------------------------------------------------
// count -> offsets -> emit inside ONE launch: after its cells are counted a wave still holds everything the emit needs (edge
// vertices in LDS, the case row and the per-cell count in registers), so the second pass's reload + recomputation and one kernel
// boundary go away.  What the waves need from each other is the canonical output offset = number of triangles of all earlier dirty
// voxels: a decoupled look-back over groups of 4 voxels (one workgroup per group).  status[g] packs a 2-bit state and
// a 30-bit value in one word, so there is no payload to order against the flag; it is published and polled with device-scope atomic
// read-modify-writes only (the same coherence every other counter in this library relies on).
// Which group a workgroup takes: with at most one group per workgroup (a stream frame: ~190 groups) simply its own index — workgroups
// are dispatched in index order, so every predecessor is running or done (mc_onepass_direct).  Otherwise groups are CLAIMED (mc_onepass_ring):
// one ticket counter per XCD hands out that XCD's runs of consecutive groups, lowest first (run q belongs to XCD q mod 8: consecutive groups are
// z-, then y-neighbours, whose 27-neighbourhoods overlap — what one of them has pulled into the XCD's L2 the next ones find there).  The groups
// nobody has claimed are then exactly the ones at and beyond the eight counters, so a workgroup that waits for such a group (its XCD lags, or has
// no workgroup resident) can take the lowest of them itself — by advancing that XCD's counter with a compare-and-swap.  Whatever the residency of
// the grid, a look-back therefore only ever waits for groups that running workgroups have claimed — no assumption about how many workgroups of the
// grid are resident, or about what else shares the GPU.
// A poll that does not succeed within MC_SPIN_LIMIT rounds gives up with DIF_C_OVERFLOW = 7 instead of hanging the queue.
#define MC_ST_AGG 0x40000000u
#define MC_ST_PREFIX 0x80000000u
#define MC_ST_VALUE 0x3FFFFFFFu
#define MC_SPIN_LIMIT (1 << 22)
#ifndef MC_RUN_MAX
#define MC_RUN_MAX 128                  /* groups per XCD run at most (512 voxels: four z-rows of a 128^3 grid) */
#endif
#ifndef MC_PATIENCE
#define MC_PATIENCE 48                  /* polls a blocked look-back waits before it looks for an unclaimed group to take */
#endif
#define MC_TICKET_STRIDE 32             /* words between the XCDs' ticket counters (a 128-byte line each) */
#define MC_TICKET_WORDS (8 * MC_TICKET_STRIDE)
__device__ __forceinline__ bool mc_published(unsigned st) { return (st & ~MC_ST_VALUE) != 0u; }
// ticket n of XCD x -> group: the XCD's runs are runs x, x + 8, x + 16, ... of `run` consecutive groups each (monotone in n)
__device__ __forceinline__ long long mc_group_of_ticket(unsigned n, int x, int run) { return ((long long)(n / (unsigned)run) * 8 + x) * run + (n % (unsigned)run); }
__device__ __forceinline__ int mc_run_length(int n_groups) {      // MC_RUN_MAX, shorter for few groups (every XCD should see several runs); a power of two
    int run = MC_RUN_MAX;
    while (run > 4 && run * 32 > n_groups) run >>= 1;
    return run;
}
__device__ __forceinline__ int mc_xcc_id() { return __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7; }       // HW_REG_XCC_ID[3:0]
// LDS of a wave: MC_RING sets of blended corners (ticket mode: groups that are counted and wait for their prefix; the direct mode uses the
// first), the 27 neighbour batches, the edge vertices of the cells being evaluated
#ifndef MC_RING
#define MC_RING 4
#endif
#ifndef MC_WAVES_PER_SIMD
#define MC_WAVES_PER_SIMD 5
#endif
#ifndef MC_XCD_RUN
#define MC_XCD_RUN 16          /* 1: group = workgroup index */
#endif
#define MC_CORNER_FLOATS(nc) ((2 * (nc) + 3) & ~3)
#define MC_RING_WAVE_LDS_FLOATS(nc) (MC_RING * MC_CORNER_FLOATS(nc) + 32 + 12 * 64)              /* ticket mode: corner ring | neighbours | edge weights */
#define MC_DIRECT_WAVE_LDS_FLOATS(nc) (MC_CORNER_FLOATS(nc) + 2 * 12 * 64)                           /* a group per workgroup: corners | edge weights | edge stds */
#define MC_ONEPASS_WAVE_LDS_FLOATS(nc) (MC_RING_WAVE_LDS_FLOATS(nc) > MC_DIRECT_WAVE_LDS_FLOATS(nc) ? MC_RING_WAVE_LDS_FLOATS(nc) : MC_DIRECT_WAVE_LDS_FLOATS(nc))

struct McVoxel {                 // what a wave knows about its voxel (wave-uniform)
    int64_t vb; int bx, by, bz;  // linear id and coordinates
    int slot, old_n, old_s;      // map slot, the voxel's previous triangle batch in the log
};

__device__ __forceinline__ void mc_voxel_coords(const McArgs& a, McVoxel& v) {
    v.bx = (int)((v.vb / ((int64_t)a.ny * a.nz)) % a.nx); v.by = (int)((v.vb / a.nz) % a.ny); v.bz = (int)(v.vb % a.nz);
}

// Dirty voxel k of the call: look-ups, the (r+1)^3 blended corners into c_sdf / c_std; returns whether the corners have both signs.
template <int RC>
__device__ __forceinline__ bool mc_load_voxel(const McArgs& a, int k, int lane, int r, float* __restrict__ c_sdf, float* __restrict__ c_std,
                                              int* __restrict__ nb, McVoxel& v) {
    const int r1 = r + 1, nc = r1 * r1 * r1;
    v.vb = a.valid_blocks[k];
    mc_voxel_coords(a, v);
    if (lane < 27) nb[lane] = mc_batch_of(a, v.bx + lane / 9 - 1, v.by + (lane / 3) % 3 - 1, v.bz + lane % 3 - 1);
    // the voxel's previous triangle batch (needed only by the emit, two dependent look-ups): requested here, beside the neighbour look-ups
    v.slot = (int)a.indexer[v.vb];
    v.old_n = a.tri_n[v.slot]; v.old_s = a.tri_start[v.slot];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    bool any_neg = false, any_pos = false;
    for (int c = lane; c < nc; c += 64) {
        float sv, dv;
        const bool ok = mc_corner(a, nb, r, c / (r1 * r1), (c / r1) % r1, c % r1, sv, dv);
        c_sdf[c] = ok ? sv : __builtin_nanf("");
        c_std[c] = ok ? dv : 0.0f;
        any_neg |= ok && sv < 0.0f;
        any_pos |= ok && !(sv < 0.0f);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    return __ballot(any_neg) != 0ull && __ballot(any_pos) != 0ull;
}

// The voxel's triangles written at `voxel_offset` among this call's triangles (lane = cell: ntri of them, `incl` = inclusive wave scan of ntri,
// edge vertices in vl), the mesh-cache log's bookkeeping for the voxel (map.py:708-709: its previous batch dies, it points at the new one).
__device__ __forceinline__ void mc_emit_voxel(const McArgs& a, int lane, int r, const McVoxel& v, int voxel_total, int voxel_offset, int ntri, int incl,
                                              unsigned long long tri_row, unsigned ok, const float* __restrict__ vl, const float* __restrict__ c_std, int64_t log_n,
                                              const float* __restrict__ vsd = nullptr) {
    for (int j = lane; j < v.old_n; j += 64) a.tri_alive[v.old_s + j] = 0;
    int64_t n_new = voxel_total;
    if (voxel_offset + n_new > a.new_limit) n_new = a.new_limit > voxel_offset ? a.new_limit - voxel_offset : 0;      // truncated by max_n_triangles
    if (log_n + voxel_offset + n_new > a.max_triangles) n_new = a.max_triangles > log_n + voxel_offset ? a.max_triangles - (log_n + voxel_offset) : 0;
    if (lane == 0) {
        a.tri_start[v.slot] = (int)(log_n + voxel_offset);
        a.tri_n[v.slot] = (int)n_new;
        if (v.old_n) atomicAdd(a.log_counters + DIF_C_CACHE_DEAD, v.old_n);
    }
    if (ntri <= 0) return;
    unsigned tl = (unsigned)(voxel_offset + (incl - ntri));                 // index among this call's triangles (< 2^30: the look-back word)
    unsigned t = tl + (unsigned)log_n;                                      // index in the log (log_n < cache_capacity < 2^31)
    const CellBox box = mc_cell_box(v.bx, v.by, v.bz, lane / (r * r), (lane / r) % r, lane % r, r);
    for (unsigned long long t3 = tri_row; (t3 & 0xF) != 0xF; t3 >>= 12) {
        const int e0 = (int)(t3 & 0xF), e1 = (int)((t3 >> 4) & 0xF), e2 = (int)((t3 >> 8) & 0xF);
        if (!mc_tri_ok(ok, t3)) continue;
        if ((int64_t)tl < a.new_limit && (int64_t)t < a.max_triangles) {
            const V4 vv[3] = {mc_vertex(vl[e0 * 64], e0, box, c_std, vsd), mc_vertex(vl[e1 * 64], e1, box, c_std, vsd), mc_vertex(vl[e2 * 64], e2, box, c_std, vsd)};
#pragma unroll
            for (int vi = 0; vi < 3; ++vi) {
                float x = vv[vi].x, y = vv[vi].y, z = vv[vi].z;
                if (a.scale) { x = x * a.vs + a.bx; y = y * a.vs + a.by; z = z * a.vs + a.bz; }   // map.py:698
                a.triangles[((int64_t)t * 3 + vi) * 3 + 0] = x;
                a.triangles[((int64_t)t * 3 + vi) * 3 + 1] = y;
                a.triangles[((int64_t)t * 3 + vi) * 3 + 2] = z;
                a.tri_std[(int64_t)t * 3 + vi] = vv[vi].w;
            }
            a.tri_id[t] = v.vb;
            a.tri_alive[t] = 1;
            if (a.out_tri && (int64_t)tl < a.out_capacity) {
#pragma unroll
                for (int vi = 0; vi < 3; ++vi) {
                    float x = vv[vi].x, y = vv[vi].y, z = vv[vi].z;
                    if (a.scale) { x = x * a.vs + a.bx; y = y * a.vs + a.by; z = z * a.vs + a.bz; }
                    a.out_tri[((int64_t)tl * 3 + vi) * 3 + 0] = x;
                    a.out_tri[((int64_t)tl * 3 + vi) * 3 + 1] = y;
                    a.out_tri[((int64_t)tl * 3 + vi) * 3 + 2] = z;
                    a.out_std[(int64_t)tl * 3 + vi] = vv[vi].w;
                }
                a.out_id[tl] = v.vb;
            }
        }
        ++t; ++tl;
    }
}

// One window of the look-back: the 64 groups idx, idx-1, ... (lane = distance).  Returns 1 when the window held a group that knows its
// inclusive prefix (sum = everything from idx down to it), 0 when it held none (sum = all 64 counts: go on at idx - 64), -1 when a group
// that is needed has not published anything yet and `block` is false (nothing consumed: ask again later), -2 when `block` is true, `patience`
// is given and the window still lacks a count after that many polls (nothing consumed: the caller looks for an unclaimed group to take).
__device__ __forceinline__ int mc_lookback_window(const McArgs& a, unsigned* __restrict__ status, int idx, int lane, bool block, int& sum, int patience = 0) {
    const int i = idx - lane;
    unsigned st = (i >= 0) ? 0u : MC_ST_PREFIX;
    int spins = 0;
    unsigned long long pre;
    int p;
    while (true) {
        if (!mc_published(st)) st = atomicOr(status + i, 0u);
        pre = __ballot((st & MC_ST_PREFIX) != 0u);
        p = pre ? (__ffsll((long long)pre) - 1) : 64;                          // nearest predecessor that knows its inclusive prefix
        const unsigned long long missing = __ballot(!mc_published(st)) & (p >= 63 ? ~0ull : ((2ull << p) - 1ull));
        if (missing == 0ull) break;
        if (!block) return -1;
        ++spins;
        if (patience > 0 && spins > patience) return -2;
        if (spins > MC_SPIN_LIMIT) { if (lane == 0) a.log_counters[DIF_C_OVERFLOW] = 7; break; }
        __builtin_amdgcn_s_sleep(2);
    }
    int v = (lane <= p) ? (int)(st & MC_ST_VALUE) : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    sum = v;
    return pre ? 1 : 0;
}

// The lowest group below `below` that NOBODY has claimed — the unclaimed groups are the ones at and beyond the XCDs' ticket counters, the lowest of
// an XCD is the group of its very next ticket —, taken by advancing that XCD's counter by one with a compare-and-swap (so the taker holds exactly
// that group, like a ticket holder).  Returns the group, or -1 if there is none (everything below is claimed: being counted by running workgroups,
// which publish without waiting for anyone) or the counter moved meanwhile (the caller polls on and may ask again).
// Such a group can be completed at once: it is the lowest unclaimed group anywhere, so everything below it is counted or being counted.
__device__ __forceinline__ int mc_take_unclaimed(unsigned* __restrict__ ticket, int below, int run, int n_groups, int lane) {
    unsigned n = 0u;
    long long g = 0x7FFFFFFFll;
    if (lane < 8) {
        n = atomicOr(ticket + lane * MC_TICKET_STRIDE, 0u);
        g = mc_group_of_ticket(n, lane, run);
        if (g >= n_groups || g >= below) g = 0x7FFFFFFFll;
    }
    int best = (int)g, who = lane;
#pragma unroll
    for (int d = 4; d >= 1; d >>= 1) {
        const int ob = __shfl_xor(best, d), ow = __shfl_xor(who, d);
        if (ob < best) { best = ob; who = ow; }
    }
    best = __shfl(best, 0); who = __shfl(who, 0);
    if (best == 0x7FFFFFFF) return -1;
    const unsigned nn = (unsigned)__shfl((int)n, who);
    unsigned old = nn + 1u;
    if (lane == 0) old = atomicCAS(ticket + who * MC_TICKET_STRIDE, nn, nn + 1u);
    return (unsigned)__shfl((int)old, 0) == nn ? best : -1;
}

// RC: the resolution as a compile-time constant (0: read it from the arguments).  The index arithmetic of the corners and cells divides by
// r, r + 1 and their squares: with a run-time r each of those is a ~30-instruction integer division, several per lane and phase.
//
// One group per workgroup (group = workgroup index): count, look-back, emit — the stream's frames.
template <int RC>
__device__ __forceinline__ void mc_onepass_direct(const McArgs& a, unsigned* __restrict__ status, float* lds, int K, int n_groups, int64_t log_n) {
    __shared__ int s_cnt[DIF_BLOCK / 64];
    __shared__ int s_excl;
    const int r = RC ? RC : a.R / 2, r1 = r + 1, nc = r1 * r1 * r1, r3 = r * r * r;
    const int lane = lane_id(), wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (wave-uniform, and the compiler is told so)
    // LDS of a wave in this mode: corners | edge weights | edge stds (the ring's other slots: the stds are kept, not rebuilt at emit — with one
    // wave per SIMD every instruction of the emit is latency); the 27 neighbour batches sit where the stds go later (they are done by then)
    float* c_sdf = lds + (size_t)wid * MC_ONEPASS_WAVE_LDS_FLOATS(nc);
    float* c_std = c_sdf + nc;
    float* vl = c_sdf + MC_CORNER_FLOATS(nc) + lane;
    float* vsd = vl + 12 * 64;
    int* nb = reinterpret_cast<int*>(c_sdf + MC_CORNER_FLOATS(nc) + 12 * 64);
    // The groups are dealt to the XCDs in runs of up to MC_XCD_RUN consecutive groups (xcd_run_item: voxels that follow each other along z,
    // then y — their 27-neighbourhoods overlap, and what one of them has pulled into the XCD's L2 the next ones find there).
    // What this costs: a group no longer waits only for LOWER workgroups (which are dispatched first, whatever the residency) — the groups in
    // front of it may sit on workgroups up to 8 MC_XCD_RUN above its own.  While a run's first groups are being counted, up to 7 (MC_XCD_RUN - 1)
    // workgroups of the other runs of the block spin in their look-back, so the launch needs 7 MC_XCD_RUN = 112 workgroups resident at a time
    // (a fifth of ONE XCD's share of this kernel); with fewer — CUs masked away from the queue — a poll runs into MC_SPIN_LIMIT and the call
    // reports DIF_C_OVERFLOW = 7 instead of hanging.  (The ticket mode below makes no such assumption.)
    if ((int)blockIdx.x >= n_groups) return;
    const int g = xcd_run_item((int)blockIdx.x, n_groups, MC_XCD_RUN);
    const int k = g * 4 + wid;
    const bool active = k < K;
    int ntri = 0;
    unsigned long long tri_row = ~0ull;
    unsigned ok = 0u;
    McVoxel v = {};
    if (active) {
        const bool crossing = mc_load_voxel<RC>(a, k, lane, r, c_sdf, c_std, nb, v);
        if (crossing && lane < r3) ntri = mc_eval_cell(a, c_sdf, c_std, vl, r, lane, tri_row, ok, vsd);
    }
    const int incl = wave_incl_scan(ntri);
    const int voxel_total = __shfl(incl, 63);
    if (lane == 0) {
        s_cnt[wid] = voxel_total;
        if (active) a.tri_count[k] = voxel_total;
    }
    __syncthreads();
    if (wid == 0) {
        const int agg = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        int excl = 0;
        if (g > 0) {
            if (lane == 0) atomicExch(status + g, MC_ST_AGG | (unsigned)agg);
            for (int idx = g - 1; idx >= 0; idx -= 64) {
                int sum;
                const int found = mc_lookback_window(a, status, idx, lane, true, sum);
                excl += sum;
                if (found) break;
            }
        }
        if (lane == 0) {
            atomicExch(status + g, MC_ST_PREFIX | ((unsigned)(excl + agg) & MC_ST_VALUE));
            s_excl = excl;
            if (g == n_groups - 1) a.log_counters[DIF_C_T] = excl + agg;            // triangles of this call (map.py:695 counts them on the host)
        }
    }
    __syncthreads();
    if (active && voxel_total > 0) {
        int voxel_offset = s_excl;
        for (int w = 0; w < wid; ++w) voxel_offset += s_cnt[w];
        mc_emit_voxel(a, lane, r, v, voxel_total, voxel_offset, ntri, incl, tri_row, ok, vl, c_std, log_n, vsd);
    }
}

// More groups than workgroups (a map with thousands of dirty voxels): groups are claimed through a ticket of the workgroup's XCD — the
// ticket holder is the only one that ever works on a group — and a workgroup does not sit on a counted group until its prefix is known: it parks the group
// (its blended corners stay in LDS, up to MC_RING - 1 sets per wave) and counts the next one; a parked group is emitted once its look-back
// succeeds (the cells are evaluated again from the parked corners: cheaper than holding their edge vertices).  The ordered commit otherwise costs
// what the slowest of the ~1,000 groups in flight in front of a group costs: measured, a third of the launch (profiles/r04_experiments.md).
// Runs per XCD: ticket n of XCD x is group ((n / R) 8 + x) R + n mod R — an XCD works through runs of R consecutive groups (R = MC_RUN_MAX, less
// for few groups), so that the cubes its voxels share with their z- and y-neighbours are fetched into ITS L2 once (every XCD took every eighth
// group until round 5: those neighbours sat in eight different L2s, and the counters showed the cubes fetched 3.3 times).  An XCD whose tickets
// are used up helps the next one.
// No deadlock, whatever the residency: a claimed group's count is published right after counting, and counting never waits.  A workgroup only
// blocks in a look-back — when its ring holds MC_RING - 1 groups or the tickets are gone — and after MC_PATIENCE polls it asks whether a group
// below its head is UNCLAIMED (its XCD lags, or has no workgroup resident): the lowest unclaimed group anywhere is the next ticket of some XCD,
// and the waiting workgroup takes exactly that ticket (mc_take_unclaimed) and completes the group at once in the ring's spare slot — everything
// below it is counted or being counted by running workgroups.  The lowest group without a count is therefore always either being counted or about
// to be taken by whoever waits for it.
template <int RC>
__device__ __forceinline__ void mc_onepass_ring(const McArgs& a, unsigned* __restrict__ status, unsigned* __restrict__ ticket, float* lds, int K,
                                                int n_groups, int64_t log_n) {
    __shared__ int s_g, s_res, s_excl, s_steal, s_hops;             // s_hops: ticket counters (this XCD's, the next one's, ...) found used up so far
    __shared__ int s_gid[MC_RING];                                   // parked groups (ring order: head .. head + cnt - 1)
    __shared__ int s_cnt[MC_RING][DIF_BLOCK / 64];                   // their voxels' triangle counts
    __shared__ int64_t s_vb[MC_RING][DIF_BLOCK / 64];                // ... and what the emit needs to know about each voxel
    __shared__ int s_slot[MC_RING][DIF_BLOCK / 64], s_oldn[MC_RING][DIF_BLOCK / 64], s_olds[MC_RING][DIF_BLOCK / 64], s_cross[MC_RING][DIF_BLOCK / 64];
    const int r = RC ? RC : a.R / 2, r1 = r + 1, nc = r1 * r1 * r1, r3 = r * r * r;
    const int lane = lane_id(), wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* ring = lds + (size_t)wid * MC_ONEPASS_WAVE_LDS_FLOATS(nc);
    int* nb = reinterpret_cast<int*>(ring + MC_RING * MC_CORNER_FLOATS(nc));
    float* vl = reinterpret_cast<float*>(nb + 32) + lane;
    if (threadIdx.x == 0) s_hops = 0;
    int head = 0, cnt = 0;
    bool exhausted = false;
    int lb_idx = -2, lb_excl = 0;                                    // wave 0: how far the oldest parked group's look-back has come (-2: not begun)

    // count group g into ring slot `slot` (all waves): corners, cells, per-voxel bookkeeping
    auto count_group = [&](int g, int slot) __attribute__((always_inline)) {
        const int k = g * 4 + wid;
        int ntri = 0;
        bool crossing = false;
        McVoxel v = {};
        if (k < K) {
            float* c_sdf = ring + slot * MC_CORNER_FLOATS(nc);
            crossing = mc_load_voxel<RC>(a, k, lane, r, c_sdf, c_sdf + nc, nb, v);
            unsigned long long tri_row;
            unsigned ok;
            if (crossing && lane < r3) ntri = mc_eval_cell(a, c_sdf, c_sdf + nc, vl, r, lane, tri_row, ok);
        }
        const int voxel_total = __shfl(wave_incl_scan(ntri), 63);
        if (lane == 0) {
            s_cnt[slot][wid] = voxel_total;
            s_vb[slot][wid] = v.vb; s_slot[slot][wid] = v.slot; s_oldn[slot][wid] = v.old_n; s_olds[slot][wid] = v.old_s; s_cross[slot][wid] = crossing;
            if (k < K) a.tri_count[k] = voxel_total;
        }
    };
    // emit the group parked in `slot` at offset s_excl (all waves): the cells again from the parked corners, the same counts
    auto emit_group = [&](int g, int slot) __attribute__((always_inline)) {
        const int k = g * 4 + wid;
        const int voxel_total = s_cnt[slot][wid];
        if (k < K && voxel_total > 0) {
            McVoxel v;
            v.vb = s_vb[slot][wid]; v.slot = s_slot[slot][wid]; v.old_n = s_oldn[slot][wid]; v.old_s = s_olds[slot][wid];
            mc_voxel_coords(a, v);
            const float* c_sdf = ring + slot * MC_CORNER_FLOATS(nc);
            int ntri = 0;
            unsigned long long tri_row = ~0ull;
            unsigned ok = 0u;
            if (s_cross[slot][wid] && lane < r3) ntri = mc_eval_cell(a, c_sdf, c_sdf + nc, vl, r, lane, tri_row, ok);
            const int incl = wave_incl_scan(ntri);
            int voxel_offset = s_excl;
            for (int w = 0; w < wid; ++w) voxel_offset += s_cnt[slot][w];
            mc_emit_voxel(a, lane, r, v, voxel_total, voxel_offset, ntri, incl, tri_row, ok, vl, c_sdf + nc, log_n);
        }
    };

    while (true) {
        const bool claim = cnt < MC_RING - 1 && !exhausted;          // (the ring's last slot stays free: a group taken over is completed there)
        if (claim && threadIdx.x == 0) {
            const int run = mc_run_length(n_groups), my_x = mc_xcc_id();
            int g = -1, hops = s_hops;
            while (hops < 8) {
                const int x = (my_x + hops) & 7;
                const long long cand = mc_group_of_ticket(atomicAdd(ticket + x * MC_TICKET_STRIDE, 1u), x, run);
                if (cand >= n_groups) { ++hops; continue; }                     // this XCD's runs are used up (its tickets only grow): help the next one
                g = (int)cand;
                break;
            }
            s_g = g;
            s_hops = hops;
        }
        __syncthreads();
        int g = -1;
        if (claim) {
            g = s_g;
            if (g < 0) exhausted = true;
        }
        const int tail = (head + cnt) % MC_RING;
        if (g >= 0) count_group(g, tail);                            // park it at the tail
        __syncthreads();
        if (wid == 0) {
            if (g >= 0) {
                const int agg = s_cnt[tail][0] + s_cnt[tail][1] + s_cnt[tail][2] + s_cnt[tail][3];
                if (lane == 0) {
                    atomicExch(status + g, MC_ST_AGG | (unsigned)agg);
                    s_gid[tail] = g;
                }
            }
            const int npend = cnt + (g >= 0 ? 1 : 0);
            int res = 0, steal = -1;
            if (npend > 0) {
                const int gh = cnt > 0 ? s_gid[head] : g;            // the oldest parked group
                const int aggh = s_cnt[head][0] + s_cnt[head][1] + s_cnt[head][2] + s_cnt[head][3];
                const bool block = npend >= MC_RING - 1 || exhausted;  // nothing else to do but wait for it
                if (lb_idx == -2) { lb_idx = gh - 1; lb_excl = 0; }
                bool done = false;
                int rounds = 0;
                while (!done) {
                    if (lb_idx < 0) { done = true; break; }
                    int sum;
                    const int found = mc_lookback_window(a, status, lb_idx, lane, block, sum, MC_PATIENCE);
                    if (found == -2) {                               // still waiting: is a group below the head nobody's?  Take the lowest.
                        if (++rounds > MC_SPIN_LIMIT / MC_PATIENCE) { if (lane == 0) a.log_counters[DIF_C_OVERFLOW] = 7; done = true; break; }
                        steal = mc_take_unclaimed(ticket, gh, mc_run_length(n_groups), n_groups, lane);
                        if (steal >= 0) break;
                        continue;                                    // (all claimed: being counted — poll on)
                    }
                    if (found < 0) break;
                    lb_excl += sum;
                    if (found) done = true; else lb_idx -= 64;
                }
                if (done) {
                    if (lane == 0) {
                        atomicExch(status + gh, MC_ST_PREFIX | ((unsigned)(lb_excl + aggh) & MC_ST_VALUE));
                        s_excl = lb_excl;
                        if (gh == n_groups - 1) a.log_counters[DIF_C_T] = lb_excl + aggh;
                    }
                    lb_idx = -2;
                    res = 1;
                }
            }
            if (lane == 0) { s_res = res; s_steal = steal; }
        }
        __syncthreads();
        if (g >= 0) ++cnt;
        if (s_res) {                                                 // the oldest parked group has its offset: emit it
            emit_group(s_gid[head], head);
            head = (head + 1) % MC_RING;
            --cnt;
        } else if (s_steal >= 0) {
            // a group the head waits for was nobody's: count it in the spare slot, look back (everything between it and the nearest prefix is
            // counted or being counted: this wait ends), emit, and go back to the head
            const int j = s_steal, spare = (head + cnt) % MC_RING;
            __syncthreads();                                         // (s_steal / s_res are rewritten below)
            count_group(j, spare);
            __syncthreads();
            if (wid == 0) {
                const int aggj = s_cnt[spare][0] + s_cnt[spare][1] + s_cnt[spare][2] + s_cnt[spare][3];
                if (lane == 0) atomicExch(status + j, MC_ST_AGG | (unsigned)aggj);
                int excl = 0;
                for (int idx = j - 1; idx >= 0; idx -= 64) {
                    int sum;
                    const int found = mc_lookback_window(a, status, idx, lane, true, sum);
                    excl += sum;
                    if (found) break;
                }
                if (lane == 0) {
                    atomicExch(status + j, MC_ST_PREFIX | ((unsigned)(excl + aggj) & MC_ST_VALUE));
                    s_excl = excl;
                    if (j == n_groups - 1) a.log_counters[DIF_C_T] = excl + aggj;
                }
            }
            __syncthreads();
            emit_group(j, spare);
            __syncthreads();
        }
        if (exhausted && cnt == 0) break;
    }
}

template <int RC>
__device__ __forceinline__ void mc_onepass_body(const McArgs& a, unsigned* __restrict__ status, unsigned* __restrict__ ticket) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int K = *a.K_ptr;
    const int n_groups = (K + 3) >> 2;
    const int64_t log_n = a.log_counters[DIF_C_CACHE_T];                 // log length before this call (k_extract_finish advances it)
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) {
            a.log_counters[DIF_C_CACHE_KEPT] = (int)log_n;
            if (K == 0) a.log_counters[DIF_C_T] = 0;
        }
        if (a.grid_tot)
            for (int t = (int)threadIdx.x; t < 1024; t += (int)blockDim.x) a.grid_tot[t] = 0;
    }
    if (n_groups > (int)gridDim.x) mc_onepass_ring<RC>(a, status, ticket, lds, K, n_groups, log_n);
    else mc_onepass_direct<RC>(a, status, lds, K, n_groups, log_n);
}

template <int RC>
__global__ void __launch_bounds__(DIF_BLOCK, RC == 4 ? MC_WAVES_PER_SIMD : 4) k_marching_cubes_onepass(McArgs a, unsigned* __restrict__ status, unsigned* __restrict__ ticket) {
    mc_onepass_body<RC>(a, status, ticket);
}
// S maps in one launch: blockIdx.y = map, each with its own look-back words and ticket (a group only ever waits for groups of ITS map that a
// running or finished workgroup has claimed, exactly as in the single launch)
struct McStream { McArgs a; unsigned* status; unsigned* ticket; };
template <int RC>
__global__ void __launch_bounds__(DIF_BLOCK, RC == 4 ? MC_WAVES_PER_SIMD : 4) k_marching_cubes_onepass_batch(Batch<McStream> b) {
    const McStream& m = b.s[blockIdx.y];
    mc_onepass_body<RC>(m.a, m.status, m.ticket);
}

// ---- a16 : device-resident mesh cache as an append-only log (map.py:703-714) -----------------------------------------
// A voxel that produced >= 1 new triangle replaces its previous batch (the reference drops cached triangles whose voxel id
// occurs among the new ones, map.py:708-709): TriScanFunctor::emit marks the old batch dead and points the voxel at its new one.

// Copies log entries [lo, lo+n) into three caller arrays with ONE launch; the destinations may be device-mapped pinned host memory
// (a streaming caller ships each frame's new triangles this way instead of three copy-engine transfers).
__global__ void __launch_bounds__(DIF_BLOCK) k_cache_export(const float* __restrict__ tri, const int64_t* __restrict__ id, const float* __restrict__ sd,
                                                          int64_t lo, int64_t n, float* __restrict__ out_tri, int64_t* __restrict__ out_id,
                                                          float* __restrict__ out_std) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t j = t0; j < n * 9; j += stride) out_tri[j] = tri[lo * 9 + j];
    for (int64_t j = t0; j < n * 3; j += stride) out_std[j] = sd[lo * 3 + j];
    for (int64_t j = t0; j < n; j += stride) out_id[j] = id[lo + j];
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

// end of extract: clear the batch map, publish the log length, hand the call's counters and (optionally) its new triangles to the caller.
// Everything here reads DIF_C_CACHE_KEPT (the log length BEFORE this call, frozen by TriScanFunctor::finish) and DIF_C_T; only the last
// statement moves DIF_C_CACHE_T, so workgroups may start in any order.
struct ExtractOut {
    int32_t* counters_out;          // [DIF_C_COUNT] or NULL
    float* tri; int64_t* id; float* sd; int64_t capacity;      // this call's new triangles (first `capacity` of them) or NULL
    int already_exported;           // the one-pass marching cubes wrote them while it emitted (or the copy is deferred)
    dif_pending_export_t* defer;    // deferred export: leave the copy to the next frame's first kernel (dif_map_t.pending_export)
    int32_t stamp; int32_t* notify; // dif_extract_buffers_t.stamp / export_notify
};

__device__ __forceinline__ void extract_finish_body(const int32_t* __restrict__ occ_slot, int32_t* __restrict__ vbm,
                                                    int* __restrict__ counters, int64_t new_limit, int64_t capacity,
                                                    const float* __restrict__ log_tri, const int64_t* __restrict__ log_id,
                                                    const float* __restrict__ log_std, const ExtractOut& out, int32_t* __restrict__ chunk_sum,
                                                    int32_t* __restrict__ super_sum, int32_t* __restrict__ dirty_tot, int n_dirty_tot, uint32_t* __restrict__ mc_status, uint32_t* __restrict__ mc_ticket,
                                                    const int* __restrict__ fc) {
    // (fc: an overlapped frame's own counter block — the integrate's counters as this frame's fusion kernel left them; the live words may be a frame
    // ahead.  K, B, VH are written by the extracts' own stream only: the live words are this frame's)
    const int B = counters[DIF_C_B];
    const int Kd = counters[DIF_C_K];
    if (mc_status)                                  // the one-pass marching cubes' look-back words of this call: back to idle 0
    {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ((Kd + 3) >> 2); i += gridDim.x * blockDim.x) mc_status[i] = 0u;
        if (blockIdx.x == 0 && threadIdx.x < 8) mc_ticket[threadIdx.x * MC_TICKET_STRIDE] = 0u;      // one ticket counter per XCD
    }
    // every dirty flag has been consumed by this call: the block totals return to idle 0 (a DEFERRED extract consumed none: k_dirty_scan)
    if (dirty_tot && counters[DIF_C_DEFERRED] == 0)
        for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n_dirty_tot; t += gridDim.x * blockDim.x) dirty_tot[t] = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) vbm[occ_slot[i]] = -1;
    if (chunk_sum)                                  // back to idle 0 (only the chunks this call's K dirty voxels could have touched)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ((Kd + 255) >> 8); i += gridDim.x * blockDim.x) {
            chunk_sum[i] = 0;
            if (i < ((Kd + 65535) >> 16)) super_sum[i] = 0;
        }
    const int64_t kept = counters[DIF_C_CACHE_KEPT];
    int64_t n_new = counters[DIF_C_T];
    if (n_new > new_limit) n_new = new_limit;
    int64_t tot = kept + n_new;
    const bool over = tot > capacity;
    if (over) { tot = capacity; n_new = capacity - kept; }
    if (out.tri && !out.already_exported) {          // destinations may be device-mapped pinned host memory: coalesced, one pass
        const int64_t n = n_new < out.capacity ? n_new : out.capacity;
        const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        for (int64_t j = t0; j < n * 9; j += stride) out.tri[j] = log_tri[kept * 9 + j];
        for (int64_t j = t0; j < n * 3; j += stride) out.sd[j] = log_std[kept * 3 + j];
        for (int64_t j = t0; j < n; j += stride) out.id[j] = log_id[kept + j];
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {      // wave 0: the call's counters with their final values, then the update itself
        const int lane = (int)threadIdx.x;
        int v = lane < DIF_C_COUNT ? counters[lane] : 0;
        const int live = v;
        // two queues: the next frame's front end may be rewriting N_OCCUPIED / ALLOC_NEW / M / C / ITEMS right now — this frame's values are the copies
        // its fusion kernel left in its own block
        if (fc && (lane == DIF_C_N_OCCUPIED || (lane >= DIF_C_ALLOC_NEW && lane <= DIF_C_ITEMS))) v = fc[DIF_FC_SHADOW + (lane == DIF_C_N_OCCUPIED ? 0 : lane - DIF_C_ALLOC_NEW + 1)];
        if (lane == DIF_C_CACHE_T) v = (int)tot;
        if (lane == DIF_C_OVERFLOW && over) v = 5;
        // (the snapshot and its stamp go to pinned host memory as write-through stores of ONE wave, the stamp after the snapshot's stores have been
        // acknowledged: a system-scope fence here wrote back and invalidated the XCD's L2 — ~4 us of this 8 us kernel)
        if (out.counters_out && lane < DIF_C_STAMP) __hip_atomic_store(out.counters_out + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (lane == 0 && out.defer) {
            const int64_t n = n_new < out.capacity ? n_new : out.capacity;
            dif_pending_export_t d;
            d.pending = n > 0 ? 1 : 0; d.kept = (int)kept; d.n = (int)n; d.seq = out.stamp;
            d.log_tri = log_tri; d.log_id = log_id; d.log_std = log_std;
            d.out_tri = out.tri; d.out_id = out.id; d.out_std = out.sd;
            d.notify = out.notify;
            *out.defer = d;
        }
        if (out.counters_out) {                      // the stamp goes out behind the snapshot: whoever sees it has all of it
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // vmcnt(0)
            if (lane == 0) __hip_atomic_store(out.counters_out + DIF_C_STAMP, out.stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (lane == DIF_C_OVERFLOW) {
            // a flag that has just been handed to the caller with this snapshot is reported: cleared here, in stream order, so that the next
            // call's snapshot neither repeats it nor loses a flag raised in between (compare-and-swap: with two queues the next frame's front end
            // may raise one at this very moment — only the value that was reported is cleared)
            if (out.counters_out) atomicCAS(counters + DIF_C_OVERFLOW, live, 0);
            else if (over) counters[DIF_C_OVERFLOW] = 5;
        }
        if (lane == 0) counters[DIF_C_CACHE_T] = (int)tot;
    }
}

struct FinishArgs {
    const int32_t* occ_slot; int32_t* vbm; int* counters; int64_t new_limit, capacity; const float* log_tri; const int64_t* log_id; const float* log_std;
    ExtractOut out; int32_t* chunk_sum; int32_t* super_sum; int32_t* dirty_tot; int n_dirty_tot; uint32_t* mc_status; uint32_t* mc_ticket;
    const int* fc;          // two queues: the frame's own counter block (dif_map_t.frame_counters) or NULL
};
__global__ void __launch_bounds__(DIF_BLOCK) k_extract_finish(FinishArgs a) {
    extract_finish_body(a.occ_slot, a.vbm, a.counters, a.new_limit, a.capacity, a.log_tri, a.log_id, a.log_std, a.out, a.chunk_sum, a.super_sum, a.dirty_tot,
                        a.n_dirty_tot, a.mc_status, a.mc_ticket, a.fc);
}
__global__ void __launch_bounds__(DIF_BLOCK) k_extract_finish_batch(Batch<FinishArgs> b) {
    const FinishArgs& a = b.s[blockIdx.y];
    extract_finish_body(a.occ_slot, a.vbm, a.counters, a.new_limit, a.capacity, a.log_tri, a.log_id, a.log_std, a.out, a.chunk_sum, a.super_sum, a.dirty_tot,
                        a.n_dirty_tot, a.mc_status, a.mc_ticket, a.fc);
}

struct TriScanFunctor {         // exclusive scan of the per-voxel triangle counts; on the mesh-cache path also the log bookkeeping
    const int32_t* tri_count;
    int32_t* tri_offset;
    int* counters;
    // mesh-cache path (valid_blocks != NULL): needs DIF_C_CACHE_KEPT frozen by the marching-cubes count pass
    const int64_t* valid_blocks; const int64_t* indexer;
    int32_t* tri_start; int32_t* tri_n; uint8_t* alive;
    int64_t new_limit, capacity;
    __device__ int count(int k) const { return tri_count[k]; }
    __device__ void emit(int k, int offset) const {          // only called for voxels with >= 1 new triangle
        tri_offset[k] = offset;
        if (!valid_blocks) return;
        const int64_t log_n = counters[DIF_C_CACHE_KEPT];
        const int64_t slot = indexer[valid_blocks[k]];
        const int old_n = tri_n[slot], old_s = tri_start[slot];
        for (int j = 0; j < old_n; ++j) alive[old_s + j] = 0;
        int64_t n_new = tri_count[k];
        if (offset + n_new > new_limit) n_new = new_limit > offset ? new_limit - offset : 0;          // truncated by max_n_triangles
        if (log_n + offset + n_new > capacity) n_new = capacity > log_n + offset ? capacity - (log_n + offset) : 0;
        tri_start[slot] = (int)(log_n + offset);
        tri_n[slot] = (int)n_new;
        // dead-entry count: one atomic per wave, summed over the lanes that are in here together
        const unsigned long long here = __ballot(1);
        int dead = 0;
        for (unsigned long long m = here; m; m &= m - 1) dead += __shfl(old_n, __ffsll((long long)m) - 1);
        if (lane_id() == __ffsll((long long)here) - 1 && dead) atomicAdd(counters + DIF_C_CACHE_DEAD, dead);
    }
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
    int32_t* count_out;          // optional (pinned host memory allowed): [0] = M, [1] = the caller's sequence number — the host slices its outputs
    int32_t seq;                 //   as soon as THIS kernel is done, while the decode of the M rows is still running
    int32_t* inv;                // [N]: row of point i among the valid ones, -1 = invalid (the inverse of sel; dif_query_grad_gather)
    __device__ int count(int i) const {
        float xn, yn, zn; int ix, iy, iz;
        bool ok = voxel_of(g, xyz[(int64_t)i * 3 + 0], xyz[(int64_t)i * 3 + 1], xyz[(int64_t)i * 3 + 2], xn, yn, zn, ix, iy, iz);
        if (ok) {
            int64_t slot = indexer[linearize(g, ix, iy, iz)];
            ok = slot >= 0 && obs[slot] > ignore_th;
        }
        mask[i] = ok ? 1 : 0;
        if (!ok) inv[i] = -1;                    // (both passes of the scan call count(): the same value twice)
        return ok ? 1 : 0;
    }
    __device__ void emit(int i, int offset) const { sel[offset] = i; inv[i] = offset; }
    __device__ void finish(int total) const {
        counters[DIF_C_QUERY_M] = total;
        if (count_out) {                         // M first, the sequence number behind a system-scope fence: a host that polls for its seq has M
            __hip_atomic_store(count_out, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // vmcnt(0): M has been acknowledged (a system-scope fence costs ~4 us of L2 write-back here)
            __hip_atomic_store(count_out + 1, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
};

// d loss / d xyz of get_sdf for the caller's autograd: out[sel[m]] = grad[m] * g_sdf[m] (out zeroed by the caller; rows of invalid points stay 0)
__global__ void __launch_bounds__(DIF_BLOCK) k_query_grad_scatter(const float* __restrict__ grad, const float* __restrict__ g_sdf, const int32_t* __restrict__ sel,
                                                                int64_t M, float* __restrict__ out) {
    NO_PACKED_F32
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < M * 3; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = e / 3;
        out[(int64_t)sel[m] * 3 + (e - m * 3)] = grad[e] * g_sdf[m];
    }
}

// The same over ALL N points through the inverse map (no zero fill of `out` needed: one launch is the whole backward)
__global__ void __launch_bounds__(DIF_BLOCK) k_query_grad_gather(const float* __restrict__ grad, const float* __restrict__ g_sdf, const int32_t* __restrict__ inv,
                                                               int64_t N, float* __restrict__ out) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < N * 3; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / 3;
        const int m = inv[i];
        out[e] = m >= 0 ? grad[(int64_t)m * 3 + (e - i * 3)] * g_sdf[m] : 0.0f;
    }
}

// =================================================================================================================
// multi-GPU merge helpers
// =================================================================================================================
struct ExportFunctor {       // ordered compaction over slots: allocated voxels with x index in [x_lo, x_hi)
    const int64_t* pos; const float* obs; const float* latent; const uint8_t* dirty;
    int32_t* rec; int64_t max_records;
    int64_t lin_lo, lin_hi;
    int raw;
    int* counters;
    int32_t* header;             // optional: record-count word that travels with the records (device-side variable length)
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
        if (header) header[0] = total;
    }
};

// Delta halo messages (dif_export_halo_delta): one 32-lane group per entry of the boundary change lists writes the voxel's raw record
// (lin | flags | w | z[29]) — both neighbours' messages in one launch.  The workgroup that takes the last ticket has seen every other
// workgroup read the list lengths (a workgroup takes its ticket after its loops, whose bounds are those lengths), so it can empty the
// lists and write the headers without a second launch.
__global__ void __launch_bounds__(DIF_BLOCK) k_export_halo_delta(HaloLists hl, const int64_t* __restrict__ pos, const float* __restrict__ obs,
                                                               const float* __restrict__ latent, const uint8_t* __restrict__ dirty,
                                                               int32_t* __restrict__ msg_l, int32_t* __restrict__ msg_r, int64_t max_records,
                                                               int* __restrict__ counters, int32_t* __restrict__ note) {
    const int grp = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 5), ngrp = (int)((gridDim.x * blockDim.x) >> 5);
    const int f = threadIdx.x & 31;
    int pending[2], n[2];
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        int32_t* msg = side ? msg_r : msg_l;
        pending[side] = counters[DIF_C_HALO_L + side];
        int64_t m = pending[side];
        if (m > hl.cap) m = hl.cap;
        if (m > max_records) m = max_records;
        n[side] = msg ? (int)m : 0;
        for (int e = grp; e < n[side]; e += ngrp) {
            const int s = hl.list[side * hl.cap + e];
            int32_t word;
            if (f == 0) word = (int32_t)pos[s];
            else if (f == 1) word = dirty[s] ? 1 : 0;
            else if (f == 2) word = __float_as_int(obs[s]);
            else word = __float_as_int(latent[(int64_t)s * L + (f - 3)]);
            msg[(int64_t)(1 + e) * 32 + f] = word;
        }
    }
    __shared__ int s_last;
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(counters + DIF_C_HALO_TICKET, 1) == (int)gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (!s_last || threadIdx.x != 0) return;
    counters[DIF_C_HALO_TICKET] = 0;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        int32_t* msg = side ? msg_r : msg_l;
        if (!msg) continue;                              // a side without a message keeps its list (a whole-layer export may follow)
        counters[DIF_C_HALO_L + side] = 0;
        msg[0] = n[side]; msg[1] = pending[side]; msg[2] = 1;
        if (note) { note[4 * side] = n[side]; note[4 * side + 1] = pending[side]; note[4 * side + 2] = 1; }
        if (pending[side] > n[side]) counters[DIF_C_OVERFLOW] = 8;
    }
}

__global__ void k_halo_lists_reset(int* __restrict__ counters, int32_t* __restrict__ hdr_l, int32_t* __restrict__ hdr_r, int32_t* __restrict__ note) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        int32_t* hdr = side ? hdr_r : hdr_l;
        if (!hdr) continue;
        const int pending = counters[DIF_C_HALO_L + side];
        hdr[1] = pending; hdr[2] = 0;
        if (note) { note[4 * side] = hdr[0]; note[4 * side + 1] = pending; note[4 * side + 2] = 0; }
        counters[DIF_C_HALO_L + side] = 0;
    }
}

// Up to two record sources per merge (the halo messages of both neighbours in one pass); record ids are distinct across both.
struct MergeSrc {
    const int32_t* rec[2]; int64_t n_static[2]; const int32_t* n_ptr[2];
    __device__ __forceinline__ int64_t count(int k) const {       // a device-side count is bounded by the buffer's capacity
        if (!rec[k]) return 0;
        return n_ptr[k] ? min((int64_t)max(*n_ptr[k], 0), n_static[k]) : n_static[k];
    }
};

__global__ void __launch_bounds__(DIF_BLOCK) k_merge_mark(MergeSrc src, const int64_t* __restrict__ indexer, GridMarks marks, int64_t grid) {
    const int64_t n0 = src.count(0), n = n0 + src.count(1);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t* rec = i < n0 ? src.rec[0] + i * 32 : src.rec[1] + (i - n0) * 32;
        int64_t lin = rec[0];
        if (lin < 0 || lin >= grid) continue;
        if (indexer[lin] == -1 && !((marks.bits[lin >> 5] >> (lin & 31)) & 1u)) marks.set((int)lin);
    }
}

// records of one call carry distinct lin ids => plain read-modify-write, deterministic
__global__ void __launch_bounds__(DIF_BLOCK) k_merge_apply(MergeSrc src, const int64_t* __restrict__ indexer, float* __restrict__ latent,
                                                         float* __restrict__ obs, uint8_t* __restrict__ dirty, int* __restrict__ counters, int64_t grid,
                                                         int64_t capacity, int assign, int* __restrict__ grid_tot, int32_t* __restrict__ note) {
    const int64_t n0 = src.count(0), n = n0 + src.count(1);
    if (note && blockIdx.x == 0 && threadIdx.x < 8) {        // the received headers, for the host's bookkeeping (may be pinned host memory)
        const int k = (int)threadIdx.x >> 2;
        note[threadIdx.x] = src.n_ptr[k] ? src.n_ptr[k][threadIdx.x & 3] : 0;
    }
    if (blockIdx.x == 0)                                     // the allocation scan has consumed the bitmap's block totals
        for (int t = (int)threadIdx.x; t < 1024; t += DIF_BLOCK) grid_tot[t] = 0;
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
        const int32_t* rec = i < n0 ? src.rec[0] + i * 32 : src.rec[1] + (i - n0) * 32;
        int64_t lin = rec[0];
        if (lin < 0 || lin >= grid) continue;
        int64_t s = indexer[lin];
        if (s < 0) continue;
        float w_r = __int_as_float(rec[2]);
        float w_old = obs[s];
        float w_new = assign ? w_r : w_old + w_r;
        if (f < L) {
            float pay = __int_as_float(rec[3 + f]);
            float z = latent[s * L + f];
            if (assign) latent[s * L + f] = pay;
            else if (w_r > 0.0f) latent[s * L + f] = (z * w_old + pay) / w_new;      // a weight-0 record only allocates (allocate_block): no re-rounding of z
        }
        __builtin_amdgcn_wave_barrier();
        if (f == 31) {
            obs[s] = w_new;
            if (assign) dirty[s] = (uint8_t)(rec[1] & 1);
            else if (w_r > 0.0f) dirty[s] = 1;
        }
    }
}
