// The fold-constant loop of the lattice decoder (mlp.hip.h:decoder_fold_consts_at) on its own, compiled WITH the SLP vectoriser (hipcc's default:
// v_pk_fma_f32 over (a0, a1) and (a2, a3), the 29 latent entries fetched as dwordx4 / dwordx2 / dword), beside waves that keep the matrix pipe busy.
// Every wave folds `iters` voxels; the result is compared with the same sum in scalar v_fma_f32 (inline asm: not vectorisable).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/pk_fma_fold.hip -o tools/micro/pk_fma_fold && tools/micro/pk_fma_fold 6000
// Measured on MI355X (ROCm 7.2, hipcc 7.2.26015; four boxes; one record in profiles/r06_pk_fma_hazard.txt):
//   * the compiled loop (fifteen loads in flight, each pair of packed FMAs right behind the counted s_waitcnt vmcnt(n) that releases its operands):
//     0 wrong folds of 12 M with no MFMA wave on the CU, 0-2 of 4-6 M with four, 5-13 of 2-3 M with six, 7-1,180 of 1-1.5 M with seven (it varies
//     from run to run); ALWAYS lanes 48..63, ALWAYS the low register of a destination pair (a0 or a2); in most failing lanes exactly ONE accumulate
//     step is missing (nearly always step k = 1), in the rest more than one: a packed FMA's write to that register never landed;
//   * the same loop with every load settled first (s_waitcnt vmcnt(0), then the 58 packed FMAs): 0;  the same loop compiled with
//     -fno-slp-vectorize (v_fma_f32, same loads, same counted waits): 0;  packed FMAs behind a wait that leaves nothing outstanding (the asm
//     variants, whose loads hipcc serialises): 0;  chains of packed FMAs on registers that settled long ago, loads in flight into OTHER registers
//     or not (tools/micro/pk_fma_hazard.hip): 0.
// So: v_pk_fma_f32 + other VMEM returns of the same wave still landing + a CU whose matrix pipes are saturated.  libdifusion holds no packed fp32
// arithmetic (csrc/common.hip.h:NO_PACKED_F32, tests/test_abi.py).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));

// MODE 0: as in the product (UNROLL loads in flight, each consumed as soon as s_waitcnt vmcnt(n) lets it through);
// MODE 1: all 58 loads first, s_waitcnt vmcnt(0), THEN the 29 steps (the packed FMAs only ever read data that landed long ago)
template <int UNROLL, int MODE>
__device__ __forceinline__ void fold_consts(const float* __restrict__ W /* LDS */, const float* __restrict__ fold, const float* __restrict__ lat_row, float* __restrict__ c, int lane) {
    float a0 = W[lane], a1 = W[lane + 64], a2 = W[128 + lane], a3 = W[128 + lane + 64];
    const f4v* wk = reinterpret_cast<const f4v*>(fold) + lane;
    if (MODE == 1) {
        float z[29]; f4v w[29];
#pragma unroll
        for (int k = 0; k < 29; ++k) { z[k] = lat_row[k]; w[k] = wk[k * 64]; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 29; ++k) {
            a0 = fmaf(w[k].x, z[k], a0); a1 = fmaf(w[k].y, z[k], a1); a2 = fmaf(w[k].z, z[k], a2); a3 = fmaf(w[k].w, z[k], a3);
        }
    } else if (MODE >= 2) {
        // the packed FMAs written out (same loads, same counted waits: hipcc puts the s_waitcnt vmcnt(n) in front of the asm statement that
        // reads a loaded register), MODE 2 as the vectoriser emits them, MODE 3 / 4 / 5 with s_nop 0 / 1 / 3 between the wait and the first read
        f2v a01 = {a0, a1}, a23 = {a2, a3};
#pragma unroll UNROLL
        for (int k = 0; k < 29; ++k) {
            const f2v z2 = {lat_row[k], 0.0f};
            const f4v wv = wk[k * 64];
            const f2v wxy = {wv.x, wv.y}, wzw = {wv.z, wv.w};
            if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %2, %4, %0 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %1, %3, %4, %1 op_sel_hi:[1,0,1]" : "+v"(a01), "+v"(a23) : "v"(wxy), "v"(wzw), "v"(z2));
            if (MODE == 3) asm volatile("s_nop 0\n\tv_pk_fma_f32 %0, %2, %4, %0 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %1, %3, %4, %1 op_sel_hi:[1,0,1]" : "+v"(a01), "+v"(a23) : "v"(wxy), "v"(wzw), "v"(z2));
            if (MODE == 4) asm volatile("s_nop 1\n\tv_pk_fma_f32 %0, %2, %4, %0 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %1, %3, %4, %1 op_sel_hi:[1,0,1]" : "+v"(a01), "+v"(a23) : "v"(wxy), "v"(wzw), "v"(z2));
            if (MODE == 5) asm volatile("s_nop 3\n\tv_pk_fma_f32 %0, %2, %4, %0 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %1, %3, %4, %1 op_sel_hi:[1,0,1]" : "+v"(a01), "+v"(a23) : "v"(wxy), "v"(wzw), "v"(z2));
            if (MODE == 6) asm volatile("v_fma_f32 %0, %4, %8, %0\n\tv_fma_f32 %1, %5, %8, %1\n\tv_fma_f32 %2, %6, %8, %2\n\tv_fma_f32 %3, %7, %8, %3" : "+v"(a01.x), "+v"(a01.y), "+v"(a23.x), "+v"(a23.y) : "v"(wv.x), "v"(wv.y), "v"(wv.z), "v"(wv.w), "v"(z2.x));
        }
        a0 = a01.x; a1 = a01.y; a2 = a23.x; a3 = a23.y;
    } else {
#pragma unroll UNROLL
        for (int k = 0; k < 29; ++k) {
            const float zk = lat_row[k];
            const f4v wv = wk[k * 64];
            a0 = fmaf(wv.x, zk, a0);
            a1 = fmaf(wv.y, zk, a1);
            a2 = fmaf(wv.z, zk, a2);
            a3 = fmaf(wv.w, zk, a3);
        }
    }
    c[lane] = a0; c[lane + 64] = a1; c[128 + lane] = a2; c[128 + lane + 64] = a3;
}
__device__ __forceinline__ float fma_asm(float a, float b, float c) {
    float d;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

template <int MODE>
__global__ void __launch_bounds__(512) k(const float* __restrict__ bias, const float* __restrict__ fold, const float* __restrict__ lat, int n_vox, int iters, int tenants,
                                         unsigned long long* __restrict__ bad /* [4 quarters][4 comps] + total + voxels */, float* __restrict__ sink) {
    __shared__ float W[256];
    __shared__ float cs[8][256];
    if (threadIdx.x < 256) W[threadIdx.x] = bias[threadIdx.x];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= 8 - tenants) {
        f16v acc;
        for (int j = 0; j < 16; ++j) acc[j] = (float)j;
        bf8v a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * j); b[j] = (__bf16)(0.002f * j); }
        for (int it = 0; it < iters * 40; ++it) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        float s = 0.f;
        for (int j = 0; j < 16; ++j) s += acc[j];
        sink[blockIdx.x * 512 + threadIdx.x] = s;
        return;
    }
    unsigned long long nb[4] = {0, 0, 0, 0}, nv = 0;
    for (int it = 0; it < iters; ++it) {
        const int v = (int)(((size_t)blockIdx.x * 8 + wave + (size_t)it * 2048) % (size_t)n_vox);
        const float* lr = lat + (size_t)v * 29;
        fold_consts<15, MODE>(W, fold, lr, cs[wave], lane);
        // the same sums, one scalar FMA at a time
        float r0 = W[lane], r1 = W[lane + 64], r2 = W[128 + lane], r3 = W[128 + lane + 64];
        const f4v* wk = reinterpret_cast<const f4v*>(fold) + lane;
        for (int kk = 0; kk < 29; ++kk) {
            const float zk = __builtin_nontemporal_load(lr + kk);
            const f4v wv = wk[kk * 64];
            r0 = fma_asm(wv.x, zk, r0); r1 = fma_asm(wv.y, zk, r1); r2 = fma_asm(wv.z, zk, r2); r3 = fma_asm(wv.w, zk, r3);
        }
        const int b0 = cs[wave][lane] != r0, b1 = cs[wave][lane + 64] != r1, b2 = cs[wave][128 + lane] != r2, b3 = cs[wave][192 + lane] != r3;
        nb[0] += b0; nb[1] += b1; nb[2] += b2; nb[3] += b3;
        if (b0 | b2) {          // which term is it?  the k whose product w[k] * z[k] is nearest to (right - wrong), and how near
            const float diff = b0 ? r0 - cs[wave][lane] : r2 - cs[wave][128 + lane];
            int best = -1; float br = 1e30f;
            for (int kk = 0; kk < 29; ++kk) {
                const f4v wv = wk[kk * 64];
                const float term = (b0 ? wv.x : wv.z) * lr[kk];
                if (fabsf(diff - term) < br) { br = fabsf(diff - term); best = kk; }
            }
            atomicAdd(bad + 18 + best, 1ull);
            if (br > 1e-6f * fmaxf(1.0f, fabsf(diff) * 16.f)) atomicAdd(bad + 18 + 29, 1ull);       // not a single missing term
        }
        nv += __ballot(b0 | b1 | b2 | b3) != 0 && lane == 0;
    }
    for (int cidx = 0; cidx < 4; ++cidx)
        if (nb[cidx]) { atomicAdd(bad + (lane >> 4) * 4 + cidx, nb[cidx]); atomicAdd(bad + 16, nb[cidx]); }
    if (nv) atomicAdd(bad + 17, nv);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000, blocks = 256, n_vox = 12765;
    std::vector<float> hb(256), hf(29 * 256), hl((size_t)n_vox * 29);
    srand(1);
    for (auto& x : hb) x = (rand() / (float)RAND_MAX - 0.5f);
    for (auto& x : hf) x = (rand() / (float)RAND_MAX - 0.5f) * 0.4f;
    for (auto& x : hl) x = (rand() / (float)RAND_MAX - 0.5f) * 0.6f;
    float *b, *f, *l, *sink; unsigned long long* bad;
    hipMalloc(&b, 1024); hipMalloc(&f, 29 * 1024); hipMalloc(&l, hl.size() * 4); hipMalloc(&sink, blocks * 512 * 4); hipMalloc(&bad, 48 * 8);
    hipMemcpy(b, hb.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(f, hf.data(), 29 * 1024, hipMemcpyHostToDevice); hipMemcpy(l, hl.data(), hl.size() * 4, hipMemcpyHostToDevice);
    const char* names[7] = {"compiled: loads consumed as they land", "compiled: all loads settled first (s_waitcnt vmcnt(0))", "asm v_pk_fma_f32 right behind the counted wait",
                            "asm, s_nop 0 behind the wait", "asm, s_nop 1 behind the wait", "asm, s_nop 3 behind the wait", "asm, four v_fma_f32 behind the wait"};
    for (int mode = 0; mode < 7; ++mode)
    for (int tenants : {0, 4, 6, 7}) {
        hipMemset(bad, 0, 48 * 8);
        void (*kern[7])(const float*, const float*, const float*, int, int, int, unsigned long long*, float*) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>};
        hipLaunchKernelGGL(kern[mode], dim3(blocks), dim3(512), 0, 0, b, f, l, n_vox, iters, tenants, bad, sink);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        unsigned long long h[48]; hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost);
        printf("[%s] %d MFMA waves beside %d folding waves per CU: %llu wrong constants in %llu of %.3g folds; by lane quarter x (a0 a1 a2 a3):", names[mode], tenants, 8 - tenants, h[16], h[17],
               (double)blocks * (8 - tenants) * iters);
        for (int q = 0; q < 4; ++q) printf("  q%d [%llu %llu %llu %llu]", q, h[q * 4], h[q * 4 + 1], h[q * 4 + 2], h[q * 4 + 3]);
        printf("\n");
        if (h[16]) {
            printf("      the missing term's k (count):");
            for (int kk = 0; kk < 29; ++kk) if (h[18 + kk]) printf(" %d(%llu)", kk, h[18 + kk]);
            printf("; lanes whose error is NOT one missing term: %llu\n", h[18 + 29]);
        }
    }
    return 0;
}
