// 8f-4: latent optimisation of confident voxels (reference system/map.py:459-513 + OptimizeProcess.do_optimize :80-113 +
// _update_optimize_result_set(deintegrate_old=False) :321-335).  (part of libdifusion; included by difusion.hip inside its anonymous namespace)
//
// Stage 3 of integrate_keyframe(do_optimize=True): the voxels that have left the encoder's care (observation count >= encoder_count_th)
// and were not optimised before get their latent refined by a few Adam steps on the negative log likelihood of perturbed surface
// samples under the decoder's (sdf, std) prediction.  Everything runs on the device, enqueued on one stream: gather (ordered, so that
// row k receives the caller's k-th perturbation sample exactly as the reference's torch.randn stream would be consumed), the unique
// voxel list, n_iters x (decoder forward + reverse MFMA chain -> per-voxel gradient sums in fixed point) + Adam, write-back.
#pragma once

struct OptimSet {               // "in the optimisation set" (map.py:465-467)
    const float* obs; const uint8_t* optimized; const int64_t* pos; float enc_th;
    __device__ __forceinline__ bool has(int64_t slot) const { return slot >= 0 && obs[slot] >= enc_th && !optimized[slot] && pos[slot] > 0; }
};

// focus[i] = point i survived the pruning AND its own voxel or an in-grid 6-neighbour is in the set (get_pruned_surface, map.py:389-399)
__global__ void __launch_bounds__(DIF_BLOCK) k_optim_focus(Geo g, OptimSet S, const float* __restrict__ xyz, const uint8_t* __restrict__ unq_mask, int64_t N,
                                                         const int64_t* __restrict__ indexer, uint8_t* __restrict__ focus, int* __restrict__ counters) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { counters[DIF_C_OPT_ROWS] = 0; counters[DIF_C_OPT_VOXELS] = 0; }
    if (i >= N) return;
    float xn, yn, zn; int ix, iy, iz;
    bool f = false;
    if (unq_mask[i] && voxel_of(g, xyz[i * 3 + 0], xyz[i * 3 + 1], xyz[i * 3 + 2], xn, yn, zn, ix, iy, iz)) {
        const int lin = linearize(g, ix, iy, iz), plane = g.ny * g.nz;
        const int nl[7] = {lin, lin - plane, lin + plane, lin - g.nz, lin + g.nz, lin - 1, lin + 1};
        const bool in_grid[7] = {true, ix > 0, ix < g.nx - 1, iy > 0, iy < g.ny - 1, iz > 0, iz < g.nz - 1};
#pragma unroll
        for (int q = 0; q < 7; ++q) f |= in_grid[q] && S.has(indexer[nl[q]]);
    }
    focus[i] = f ? 1 : 0;
}

// Ordered compaction over the pairs j = o*N + i (the reference's concatenation order, map.py:479-497): row k = k-th valid pair.
struct OptimGatherFunctor {
    Geo g; OptimSet S;
    const float* xyz; const float* normal; const uint8_t* focus; int64_t N;
    const int64_t* indexer; const float* noise;
    int* row_slot; float* row_xyz; float* row_sdf; int* slot_flag; int* counters;
    __device__ __forceinline__ int64_t neighbour(int j, float& rx, float& ry, float& rz, int64_t& i) const {
        const int o = (int)(j / N);
        i = j - (int64_t)o * N;
        if (!focus[i]) return -1;
        float xn, yn, zn; int ix, iy, iz;
        voxel_of(g, xyz[i * 3 + 0], xyz[i * 3 + 1], xyz[i * 3 + 2], xn, yn, zn, ix, iy, iz);
        const float ox = (o & 4) ? 0.5f : -0.5f, oy = (o & 2) ? 0.5f : -0.5f, oz = (o & 1) ? 0.5f : -0.5f;      // map.py:186-189
        const float gx = fminf(fmaxf(ceilf(xn + ox) - 1.0f, 0.0f), (float)(g.nx - 1));
        const float gy = fminf(fmaxf(ceilf(yn + oy) - 1.0f, 0.0f), (float)(g.ny - 1));
        const float gz = fminf(fmaxf(ceilf(zn + oz) - 1.0f, 0.0f), (float)(g.nz - 1));
        rx = (xn - gx) - 0.5f; ry = (yn - gy) - 0.5f; rz = (zn - gz) - 0.5f;                                      // map.py:483
        const int64_t slot = indexer[linearize(g, (int)gx, (int)gy, (int)gz)];
        return S.has(slot) ? slot : -1;
    }
    __device__ int count(int j) const { float a, b, c; int64_t i; return neighbour(j, a, b, c, i) >= 0 ? 1 : 0; }
    __device__ void emit(int j, int k) const {
        float rx, ry, rz; int64_t i;
        const int64_t slot = neighbour(j, rx, ry, rz, i);
        const float sdf = noise[k] * 0.05f;                                                                          // map.py:489
        row_slot[k] = (int)slot;
        row_xyz[(int64_t)k * 3 + 0] = rx + sdf * normal[i * 3 + 0];                                                  // map.py:490
        row_xyz[(int64_t)k * 3 + 1] = ry + sdf * normal[i * 3 + 1];
        row_xyz[(int64_t)k * 3 + 2] = rz + sdf * normal[i * 3 + 2];
        row_sdf[k] = sdf;
        slot_flag[slot] = 1;
    }
    __device__ void finish(int total) const { counters[DIF_C_OPT_ROWS] = total; }
};

// torch.unique(gathered_latent_inds) (map.py:496): ascending slots; slot_u = the inverse mapping
struct OptimUniqueFunctor {
    const int* slot_flag; int* slot_u; int* uniq_slot; int* counters;
    __device__ int count(int s) const { return slot_flag[s] ? 1 : 0; }
    __device__ void emit(int s, int u) const { uniq_slot[u] = s; slot_u[s] = u; }
    __device__ void finish(int total) const { counters[DIF_C_OPT_VOXELS] = total; }
};

#define DIF_OPT_FIX_SCALE 1099511627776.0f        /* 2^40: gradient sums are tiny (loss / n_samples); exact integer accumulation */

__global__ void __launch_bounds__(DIF_BLOCK) k_optim_init(const int* __restrict__ uniq_slot, const float* __restrict__ latent, float* __restrict__ z,
                                                        float* __restrict__ m, float* __restrict__ v, long long* __restrict__ grad,
                                                        const int* __restrict__ counters) {
    const int n = counters[DIF_C_OPT_VOXELS] * 32;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const int u = e >> 5, f = e & 31;
        z[e] = f < L ? latent[(int64_t)uniq_slot[u] * L + f] : 0.0f;
        m[e] = 0.0f; v[e] = 0.0f; grad[e] = 0;
    }
}

// One Adam iteration, part 1: rows through the decoder forward + reverse chain; d loss / d latent summed per voxel.
// 256 threads = one wave per SIMD with the full register budget (as k_decode<GRAD>).
// X6: the tile runs on the bf16 matrix pipe (decoder_tile_nll_grad_x6: wblob = pack_decoder_x6, wbwd_blob = pack_decoder_x6_backward, wu_blob =
// pack_decoder_x6u); otherwise on the f32-input MFMA (wblob = pack_decoder, wbwd_blob = pack_decoder_backward).
template <bool X6>
__global__ void __launch_bounds__(256, 1) k_optim_grad(const float* __restrict__ wblob, const float* __restrict__ wbwd_blob, const float* __restrict__ wu_blob, const int* __restrict__ row_slot,
                                                     const float* __restrict__ row_xyz, const float* __restrict__ row_sdf, const int* __restrict__ slot_u,
                                                     const float* __restrict__ z, unsigned long long* __restrict__ grad, float* __restrict__ loss_sum,
                                                     const int* __restrict__ counters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    stage_weights(lds, wblob, X6 ? X6_LDS_BYTES / 4 : DEC_LDS_FLOATS);
    const __amdgpu_buffer_rsrc_t wfwd = make_rsrc(wblob, X6 ? X6_BYTES / 4 : DEC_FLOATS), wbwd = make_rsrc(wbwd_blob, X6 ? X6B_BYTES / 4 : DECB_FLOATS);
    const __amdgpu_buffer_rsrc_t wun = make_rsrc(X6 ? wu_blob : wblob, X6 ? X6U_BYTES / 4 : 4);
    const int lane = lane_id(), half = lane >> 5, col = lane & 31;
    const int wave = (int)((threadIdx.x >> 6) * gridDim.x + blockIdx.x), nwaves = (int)(gridDim.x * (blockDim.x >> 6));
    const int n_rows = counters[DIF_C_OPT_ROWS];
    const float inv_n = 1.0f / (float)n_rows;                                      // n_samples, map.py:85
    const int n_tiles = (n_rows + 31) >> 5;
    for (int tile = wave; tile < n_tiles; tile += nwaves) {
        const int row = tile * 32 + col;
        const bool live = row < n_rows;
        const int u = live ? slot_u[row_slot[row]] : -1;
        const float* zr = z + (int64_t)(live ? u : 0) * 32;
        f16v xin;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int k = 2 * t + half;
            float val = 0.0f;
            if (live) val = k < L ? zr[k] : row_xyz[(int64_t)row * 3 + (k - L)];
            xin[t] = val;
        }
        float sdf, sd, loss;
        f16v gx;
        if constexpr (X6) decoder_tile_nll_grad_x6<GRAD_X6_PF>(lds, wfwd, wun, wbwd, xin, lane, live ? row_sdf[row] : 0.0f, inv_n, sdf, sd, loss, gx);
        else decoder_tile_nll_grad(lds, wfwd, wbwd, xin, lane, live ? row_sdf[row] : 0.0f, inv_n, sdf, sd, loss, gx);
        // per-voxel sums: segmented scan over runs of equal voxels, the last lane of a run adds the run's sum (exact fixed point, so the
        // result does not depend on the order the atomics land in)
        const int u_prev = __shfl_up(u, 1), u_next = __shfl_down(u, 1);
        const bool run_head = (col == 0) || (u_prev != u), run_tail = (col == 31) || (u_next != u);
        const uint32_t heads32 = (uint32_t)(__ballot(run_head) >> (half * 32));
        const int my_head = 31 - __clz((int)(heads32 & ((2u << col) - 1u)));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            long long q = live ? __float2ll_rn(gx[r] * DIF_OPT_FIX_SCALE) : 0ll;
            q = seg_incl_scan32(q, col, my_head);
            const int f = (r & 3) + 8 * (r >> 2) + 4 * half;
            if (live && run_tail && f < L) (void)__hip_atomic_fetch_add(grad + (int64_t)u * 32 + f, (unsigned long long)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (loss_sum) {                                                              // diagnostics: the likelihood part of the loss
            float ls = (live && half == 0) ? loss : 0.0f;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) ls += __shfl_xor(ls, d);
            if (lane == 0) atomicAdd(loss_sum, ls);
        }
    }
}

// One Adam iteration, part 2 (torch.optim.Adam defaults, map.py:83): one 32-lane group per voxel.
__global__ void __launch_bounds__(DIF_BLOCK) k_optim_adam(float* __restrict__ z, float* __restrict__ m, float* __restrict__ v, long long* __restrict__ grad,
                                                        const int* __restrict__ counters, int iter, float lr, float reg_lambda) {
    const int n_vox = counters[DIF_C_OPT_VOXELS];
    const float inv_n = 1.0f / (float)counters[DIF_C_OPT_ROWS];
    const int grp = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 5), ngrp = (int)((gridDim.x * blockDim.x) >> 5), f = threadIdx.x & 31;
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    const float step = lr / (1.0f - powf(b1, (float)iter)), bc2 = sqrtf(1.0f - powf(b2, (float)iter));
    for (int u = grp; u < n_vox; u += ngrp) {
        const int e = u * 32 + f;
        const float zf = z[e];
        float g = (float)grad[e] * (1.0f / DIF_OPT_FIX_SCALE);
        grad[e] = 0;
        if (reg_lambda > 0.0f) {                                                    // code regulariser: lambda * sum_u |z_u| / n_samples (map.py:98-101)
            float nn = f < L ? zf * zf : 0.0f;
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1) nn += __shfl_xor(nn, d, 32);
            if (nn > 0.0f) g += reg_lambda * zf / sqrtf(nn) * inv_n;          // torch.norm's backward: subgradient 0 at z = 0 (a merged / loaded all-zero latent)
        }
        if (f < L) {
            const float mf = b1 * m[e] + (1.0f - b1) * g;
            const float vf = b2 * v[e] + (1.0f - b2) * g * g;
            m[e] = mf; v[e] = vf;
            z[e] = zf - step * (mf / (sqrtf(vf) / bc2 + eps));
        }
    }
}

// _update_optimize_result_set(deintegrate_old=False) (map.py:321-335): latents written back, voxels marked optimised and dirty
__global__ void __launch_bounds__(DIF_BLOCK) k_optim_writeback(const int* __restrict__ uniq_slot, const float* __restrict__ z, float* __restrict__ latent,
                                                             uint8_t* __restrict__ optimized, uint8_t* __restrict__ dirty, int* __restrict__ slot_flag,
                                                             const int* __restrict__ counters) {
    const int n = counters[DIF_C_OPT_VOXELS] * 32;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const int u = e >> 5, f = e & 31, s = uniq_slot[u];
        if (f < L) latent[(int64_t)s * L + f] = z[e];
        if (f == 31) { optimized[s] = 1; dirty[s] = 1; slot_flag[s] = 0; }      // (a tiled map's façade refreshes its neighbours with whole-layer halo messages after this)
    }
}
