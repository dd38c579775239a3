// Does a chain of v_pk_fma_f32 on two alternating accumulator pairs (the code hipcc's SLP vectoriser makes of decoder_fold_consts_at, mlp.hip.h)
// lose a term now and then?  di_fusion_amd/_build.py builds with -fno-slp-vectorize because the lattice kernel built WITH the vectoriser returns,
// in ~10 of 12,765 voxels per launch, a fold constant row whose lanes 48..63 are one term short in the LOW half of one packed accumulator.
//   acc0 = (0, 0), acc1 = (0, 0); 29 x { acc0 += w0 * z.x ; acc1 += w1 * z.x }   with w = (1, 1), z = (1, junk)  ->  every component must be 29
// flags: bit0  the other four waves of the workgroup (one per SIMD) run v_mfma_f32_32x32x16_bf16 back to back
//        bit1  sixteen 16-byte global loads are in flight while the chain runs (their data lands in other registers)
//        bit2  s_nop 0 between the packed FMAs
//        bit3  the chain in scalar v_fma_f32 (what the product build uses)
// hipcc --offload-arch=gfx950 -O3 tools/micro/pk_fma_hazard.hip -o tools/micro/pk_fma_hazard && tools/micro/pk_fma_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));

#define STEP_PK(NOP) \
    asm volatile("v_pk_fma_f32 %0, %2, %4, %0 op_sel_hi:[1,0,1]\n\t" NOP "v_pk_fma_f32 %1, %3, %4, %1 op_sel_hi:[1,0,1]\n\t" NOP \
                 : "+v"(a0), "+v"(a1) : "v"(w0), "v"(w1), "v"(z));
#define STEP_SC \
    asm volatile("v_fma_f32 %0, %4, %8, %0\n\tv_fma_f32 %1, %5, %8, %1\n\tv_fma_f32 %2, %6, %8, %2\n\tv_fma_f32 %3, %7, %8, %3\n\t" \
                 : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(w0.x), "v"(w0.y), "v"(w1.x), "v"(w1.y), "v"(z.x));

__global__ void __launch_bounds__(512) k(const f4v* __restrict__ g, int iters, int flags, int tenants, unsigned long long* __restrict__ bad /* [4 quarters][4 components] + [16] total */,
                                         float* __restrict__ sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((flags & 1) && wave >= 8 - tenants) {           // the matrix pipe's tenants
        f16v acc;
        for (int j = 0; j < 16; ++j) acc[j] = (float)j;
        bf8v a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * j); b[j] = (__bf16)(0.002f * j); }
        for (int it = 0; it < iters * 8; ++it) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        float s = 0.f;
        for (int j = 0; j < 16; ++j) s += acc[j];
        sink[blockIdx.x * 512 + threadIdx.x] = s;
        return;
    }
    const f2v w0 = {1.0f, 1.0f}, w1 = {1.0f, 1.0f};
    f2v z = {1.0f, 123456.0f};
    unsigned long long nbad[4] = {0, 0, 0, 0};
    float keep = 0.f;
    const f4v* gp = g + ((size_t)blockIdx.x * 512 + threadIdx.x) % 4096;
    for (int it = 0; it < iters; ++it) {
        f4v t[16];
        if (flags & 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) t[i] = __builtin_nontemporal_load(gp + ((it * 16 + i) % 64) * 4096);
        }
        f2v a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (flags & 8) {
#pragma unroll
            for (int k2 = 0; k2 < 29; ++k2) STEP_SC
            a0 = f2v{s0, s1}; a1 = f2v{s2, s3};
        } else if (flags & 4) {
#pragma unroll
            for (int k2 = 0; k2 < 29; ++k2) STEP_PK("s_nop 0\n\t")
        } else {
#pragma unroll
            for (int k2 = 0; k2 < 29; ++k2) STEP_PK("")
        }
        nbad[0] += a0.x != 29.0f; nbad[1] += a0.y != 29.0f; nbad[2] += a1.x != 29.0f; nbad[3] += a1.y != 29.0f;
        if (flags & 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) keep += t[i].x;
        }
    }
    for (int c = 0; c < 4; ++c)
        if (nbad[c]) { atomicAdd(bad + (lane >> 4) * 4 + c, nbad[c]); atomicAdd(bad + 16, nbad[c]); }
    sink[blockIdx.x * 512 + threadIdx.x] = keep;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000, blocks = 256;
    f4v* g; unsigned long long* bad; float* sink;
    hipMalloc(&g, 64 * 4096 * sizeof(f4v)); hipMemset(g, 0, 64 * 4096 * sizeof(f4v));
    hipMalloc(&bad, 17 * 8); hipMalloc(&sink, blocks * 512 * 4);
    for (int tenants : {4, 6, 7})
    for (int flags : {1, 3, 4 | 1, 8 | 1}) {
        hipMemset(bad, 0, 17 * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, g, iters, flags, tenants, bad, sink);
        hipEventRecord(e1);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[17]; hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost);
        const double chains = (double)blocks * ((flags & 1) ? 8 - tenants : 8) * 64 * iters;
        printf("%d MFMA waves; flags %2d (%s%s%s%s): %llu wrong components of %.3g lane-chains, %.1f ms; by lane quarter x (acc0.lo acc0.hi acc1.lo acc1.hi):", tenants, flags,
               (flags & 1) ? "mfma tenant " : "", (flags & 2) ? "loads in flight " : "", (flags & 4) ? "s_nop " : "", (flags & 8) ? "scalar fma" : "packed fma", h[16], chains, ms);
        for (int q = 0; q < 4; ++q) printf("  q%d [%llu %llu %llu %llu]", q, h[q * 4], h[q * 4 + 1], h[q * 4 + 2], h[q * 4 + 3]);
        printf("\n");
    }
    return 0;
}
