// How fast does one SIMD retire v_mfma_f32_32x32x16_bf16 as a function of the accumulator pattern?  
//   pattern 0: one accumulator, every MFMA depends on the previous one
//   pattern 1: four accumulators round-robin (no MFMA depends on its predecessor)
//   pattern 2: six MFMAs on one accumulator, then the next accumulator (the order of step_x6 in mlp.hip.h)
//   pattern 3: pattern 2 with four independent VALU instructions between the MFMAs of a group
// Every CU runs `waves` waves (4 = one per SIMD, 8 = two per SIMD); cycles from s_memtime, time from s_memrealtime (100 MHz).
// Measured on MI355X (ROCm 7.2), 480 MFMAs per wave:
//   the shader clock under this load is 2.0-2.1 GHz, not the 2.4 GHz of the data sheet;
//   ONE wave retires an MFMA every 33.8 cycles whatever the accumulator pattern (0, 1, 2: no penalty for six dependent MFMAs in a row),
//   TWO waves on a SIMD retire one every 25.2 cycles between them;
//   four independent VALU instructions per MFMA are free (pattern 3); the three LDS fragment reads per step cost one cycle (4: 35.0 / 26.0);
//   thirty-six VALU instructions per step that depend on each other in pairs are not hidden, wherever they are put (5, 7, 8, 9:
//   53-64 cycles alone, 40-47 per SIMD with two waves): VALU work of that density is paid on top of the matrix time.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_chain.hip -o tools/micro/mfma_chain && tools/micro/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));

template <int PATTERN>
__global__ void __launch_bounds__(512) k(float* out, unsigned long long* stamps, int reps) {
    f16v acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = (float)(threadIdx.x + i + j);
    bf8v a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x & 7) + j); b[j] = (__bf16)(0.002f * j); }
    float f0 = threadIdx.x, f1 = 1.0f, f2 = 2.0f, f3 = 3.0f;
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter(), t0 = wall_clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                const int i = PATTERN == 0 ? 0 : PATTERN == 1 ? (g * 6 + m) & 3 : g;
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                if (PATTERN == 3) {
                    f0 = f0 * 1.0001f + 0.5f; f1 = f1 * 0.9999f + f0; f2 = f2 * 1.0002f + 0.25f; f3 = f3 * 0.9998f + f2;
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
            }
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
    float s = f0 + f1 + f2 + f3;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = c1 - c0;
        stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = t1 - t0;
    }
}

// patterns 4..7: the shape of layer_x6 — per step three 16-byte A fragments (hi / mid / lo slices) from LDS one step ahead, six
// MFMAs on one accumulator with the slice products of step_x6; 5: + 36 VALU instructions per step; 6: A from L2 (global) instead of LDS
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
struct Tri { u4v q0, q1, q2; };
__device__ __forceinline__ f16v mfb(u4v a, u4v b, f16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8v, a), __builtin_bit_cast(bf8v, b), c, 0, 0, 0);
}
template <int PATTERN>
__global__ void __launch_bounds__(512) k2(float* out, unsigned long long* stamps, int reps, const u4v* __restrict__ gw) {
    extern __shared__ __attribute__((aligned(16))) u4v lw[];
    for (int i = threadIdx.x; i < 24 * 192; i += blockDim.x) lw[i] = gw[i];
    f16v acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = (float)(threadIdx.x + i + j);
    Tri x;
    x.q0 = gw[threadIdx.x & 63]; x.q1 = gw[64 + (threadIdx.x & 63)]; x.q2 = gw[128 + (threadIdx.x & 63)];
    float f[4] = {(float)threadIdx.x, 1.0f, 2.0f, 3.0f};
    const int lane = threadIdx.x & 63;
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter(), t0 = wall_clock64();
    auto load = [&](int t) {
        const u4v* q = (PATTERN == 6 ? gw : lw) + (t % 24) * 192 + lane;
        return Tri{q[0], q[64], q[128]};
    };
    if (PATTERN == 8 || PATTERN == 9) {       // four accumulators round-robin inside each slice product: no MFMA follows one on its own accumulator
        Tri an[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) an[g] = load(g);
        for (int r = 0; r < reps; ++r) {
            Tri a[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) { a[g] = an[g]; an[g] = load(r * 4 + g + 4); }
#pragma unroll
            for (int v = 0; v < 36; ++v) { f[0] = f[0] * 1.0001f + 0.5f; f[1] = f[1] * 0.9999f + f[0]; f[2] = f[2] * 1.0002f + 0.25f; f[3] = f[3] * 0.9998f + f[2]; }
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = mfb(a[g].q2, x.q0, acc[g]);
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = mfb(a[g].q1, x.q1, acc[g]);
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = mfb(a[g].q0, x.q2, acc[g]);
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = mfb(a[g].q1, x.q0, acc[g]);
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = mfb(a[g].q0, x.q1, acc[g]);
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = mfb(a[g].q0, x.q0, acc[g]);
            if (PATTERN == 9) {
#pragma unroll
                for (int m = 0; m < 24; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                    if (m % 2 == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    Tri nxt = load(0);
    for (int r = 0; r < ((PATTERN == 8 || PATTERN == 9) ? 0 : reps); ++r) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const Tri a = nxt;
            nxt = load(r * 4 + g + 1);
            if (PATTERN >= 5) {
#pragma unroll
                for (int v = 0; v < 9; ++v) { f[0] = f[0] * 1.0001f + 0.5f; f[1] = f[1] * 0.9999f + f[0]; f[2] = f[2] * 1.0002f + 0.25f; f[3] = f[3] * 0.9998f + f[2]; }
            }
            acc[g] = mfb(a.q2, x.q0, acc[g]);
            acc[g] = mfb(a.q1, x.q1, acc[g]);
            acc[g] = mfb(a.q0, x.q2, acc[g]);
            acc[g] = mfb(a.q1, x.q0, acc[g]);
            acc[g] = mfb(a.q0, x.q1, acc[g]);
            acc[g] = mfb(a.q0, x.q0, acc[g]);
            if (PATTERN == 7) {             // pattern 5 with the VALU work dealt out evenly: one MFMA, six VALU, ...
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
                for (int m = 0; m < 6; ++m) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 6, 0); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
    float s = f[0] + f[1] + f[2] + f[3] + nxt.q0[0];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = c1 - c0;
        stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = t1 - t0;
    }
}

template <int P>
void run2(int waves, float* out, unsigned long long* st, unsigned long long* h, int reps, const u4v* gw) {
    hipFuncSetAttribute((const void*)k2<P>, hipFuncAttributeMaxDynamicSharedMemorySize, 24 * 192 * 16 + 80 * 1024);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k2<P>, dim3(256), dim3(64 * waves), 24 * 192 * 16 + 80 * 1024, 0, out, st, reps, gw);   // > half the LDS: one workgroup per CU
    hipDeviceSynchronize();
    hipMemcpy(h, st, 256 * 8 * 2 * 8, hipMemcpyDeviceToHost);
    double cyc = 0, ticks = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) { cyc += h[(b * 8 + w) * 2]; ticks += h[(b * 8 + w) * 2 + 1]; }
    const double n = 256.0 * waves, mf = 24.0 * reps;
    const double per_wave = cyc / n / mf, us = ticks / n / 100.0;
    printf("pattern %d, %d waves/CU: %.1f cycles per MFMA per wave, %.1f per SIMD, %.2f us per %d MFMAs, clock %.2f GHz\n", P, waves, per_wave,
           per_wave / (waves / 4.0), us, (int)mf, cyc / n / (us * 1e3));
}

template <int P>
void run(int waves, float* out, unsigned long long* st, unsigned long long* h, int reps) {
    hipLaunchKernelGGL(k<P>, dim3(256), dim3(64 * waves), 0, 0, out, st, reps);       // warm-up
    hipLaunchKernelGGL(k<P>, dim3(256), dim3(64 * waves), 0, 0, out, st, reps);
    hipDeviceSynchronize();
    hipMemcpy(h, st, 256 * 8 * 2 * 8, hipMemcpyDeviceToHost);
    double cyc = 0, ticks = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) { cyc += h[(b * 8 + w) * 2]; ticks += h[(b * 8 + w) * 2 + 1]; }
    const double n = 256.0 * waves, mf = 24.0 * reps;
    const double per_wave = cyc / n / mf, per_simd = per_wave / (waves / 4.0), us = ticks / n / 100.0;
    printf("pattern %d, %d waves/CU: %.1f cycles per MFMA per wave, %.1f per SIMD, %.2f us per %d MFMAs, clock %.2f GHz\n", P, waves, per_wave, per_simd,
           us, (int)mf, cyc / n / (us * 1e3));
}

int main() {
    float* out; unsigned long long *st, *h = new unsigned long long[256 * 8 * 2];
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&st, 256 * 8 * 2 * 8);
    for (int waves : {4, 8}) {
        run<0>(waves, out, st, h, 20); run<1>(waves, out, st, h, 20); run<2>(waves, out, st, h, 20); run<3>(waves, out, st, h, 20);
        run<2>(waves, out, st, h, 200);
    }
    u4v* gw; hipMalloc(&gw, 24 * 192 * 16); hipMemset(gw, 0x3c, 24 * 192 * 16);
    for (int waves : {4, 8}) { run2<4>(waves, out, st, h, 20, gw); run2<5>(waves, out, st, h, 20, gw); run2<6>(waves, out, st, h, 20, gw); run2<7>(waves, out, st, h, 20, gw); run2<8>(waves, out, st, h, 20, gw); run2<9>(waves, out, st, h, 20, gw); }
    return 0;
}
