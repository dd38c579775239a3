// Exact-fp32 MFMA MLPs (encoder 6-32-64-256-29, decoder 32-128-128-96(+32)-128-{1,1}) for gfx950.
//
// Design ("transposed chaining"): every layer is computed as  H_out^T = W * H_in^T  with the WEIGHTS as the MFMA A
// operand (M = output features) and the ACTIVATIONS as the B operand (N = 32 points per wave), using
// v_mfma_f32_32x32x2_f32 (exact f32, bitwise an fmaf chain; 157.3 TFLOP/s peak).  The D fragment of that MFMA holds,
// in lane l, point (l & 31) and features  f(r, l>>5) = (r&3) + 8*(r>>2) + 4*(l>>5)  for registers r = 0..15 — which is
// *already* a valid B fragment for the next layer if k-step r of the next layer is defined to contract features
// f(r,0) (lanes 0-31) and f(r,1) (lanes 32-63).  The host packs the weights in that k order
// (di_fusion_amd/network/packing.py), so activations never leave registers and never get shuffled between layers.
//
// Packed A layout per (layer, out-block mb, k-group g of 4 k-steps): 64 lanes x float4; lane l, component j holds
//   W[mb*32 + (l&31)][ kmap(4g + j, l>>5) ].   One ds_read_b128 (or global_load_dwordx4) feeds 4 MFMAs.
// Packed bias per (layer, mb): 2 halves x 16 floats, bias[mb*32 + f(r, half)] — loaded straight into the accumulator.
#pragma once
#include "common.hip.h"

namespace dif {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f16v mfma32(float a, float b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ f16v load_bias16(const float* b, int half) {
    const f4v* p = reinterpret_cast<const f4v*>(b + half * 16);
    f4v b0 = p[0], b1 = p[1], b2 = p[2], b3 = p[3];
    f16v r;
    r[0] = b0.x; r[1] = b0.y; r[2] = b0.z; r[3] = b0.w;
    r[4] = b1.x; r[5] = b1.y; r[6] = b1.z; r[7] = b1.w;
    r[8] = b2.x; r[9] = b2.y; r[10] = b2.z; r[11] = b2.w;
    r[12] = b3.x; r[13] = b3.y; r[14] = b3.z; r[15] = b3.w;
    return r;
}

__device__ __forceinline__ f16v relu16(f16v v) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = fmaxf(v[i], 0.0f);
    return v;
}

// acc += W_block (32 x 32*NB) * H_in (32*NB x 32 points).  `A` points at k-group 0 of the out-block.
// The A stream is software-pipelined by hand: the float4 for k-group t+1 is requested before the 4 MFMAs of k-group t
// (256 cycles of matrix-pipe time cover the LDS / L2 latency); sched_barrier pins that order — left alone, the
// scheduler hoists every ds_read of the fully unrolled chain to the top and spills hundreds of VGPRs.
template <int NB>
__device__ __forceinline__ f16v block_mm(const f4v* __restrict__ A, const f16v (&hin)[NB], f16v acc, int lane) {
    f4v a = A[lane];
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int t = kb * 4 + g;
            f4v an = a;
            if (t + 1 < NB * 4) an = A[(t + 1) * 64 + lane];
            acc = mfma32(a.x, hin[kb][4 * g + 0], acc);
            acc = mfma32(a.y, hin[kb][4 * g + 1], acc);
            acc = mfma32(a.z, hin[kb][4 * g + 2], acc);
            acc = mfma32(a.w, hin[kb][4 * g + 3], acc);
            a = an;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    return acc;
}

// ---- encoder -------------------------------------------------------------------------------------------------
// blob offsets in floats (must match packing.py:pack_encoder)
#define ENC_A0 0                      // MB=1 KG=1     256
#define ENC_B0 256                    // 32
#define ENC_A1 288                    // MB=2 KG=4     2048
#define ENC_B1 2336                   // 64
#define ENC_A2 2400                   // MB=8 KG=8     16384
#define ENC_B2 18784                  // 256
#define ENC_A3 19040                  // MB=1 KG=32    8192
#define ENC_B3 27232                  // 32
#define ENC_FLOATS 27264

// One 32-point tile through the encoder.  x0..x2 are this lane's B values for k-steps 0..2:
//   lanes 0-31: (rel.x, rel.z, n.y)   lanes 32-63: (rel.y, n.x, n.z)   of point (lane & 31).
// Returns the D fragment of the 29(+3 zero rows)-feature output.
__device__ __forceinline__ f16v encoder_tile(const float* __restrict__ W /* LDS */, float x0, float x1, float x2, int lane) {
    const int half = lane >> 5;
    f16v h0[1];
    {
        f16v acc = load_bias16(W + ENC_B0, half);
        f4v a = reinterpret_cast<const f4v*>(W + ENC_A0)[lane];
        acc = mfma32(a.x, x0, acc);
        acc = mfma32(a.y, x1, acc);
        acc = mfma32(a.z, x2, acc);
        h0[0] = relu16(acc);
    }
    f16v h1[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        f16v acc = load_bias16(W + ENC_B1 + mb * 32, half);
        acc = block_mm<1>(reinterpret_cast<const f4v*>(W + ENC_A1) + (mb * 4) * 64, h0, acc, lane);
        h1[mb] = relu16(acc);
    }
    f16v out = load_bias16(W + ENC_B3, half);
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
        f16v acc = load_bias16(W + ENC_B2 + mb * 32, half);
        acc = block_mm<2>(reinterpret_cast<const f4v*>(W + ENC_A2) + (mb * 8) * 64, h1, acc, lane);
        f16v h2[1];
        h2[0] = relu16(acc);
        out = block_mm<1>(reinterpret_cast<const f4v*>(W + ENC_A3) + (mb * 4) * 64, h2, out, lane);
    }
    return out;
}

// ---- decoder -------------------------------------------------------------------------------------------------
// blob offsets in floats (must match packing.py:pack_decoder).  [0, DEC_LDS_FLOATS) is staged in LDS, L3's A stays
// in global memory (L2-resident, 64 KB) because 196 KB of fp32 weights do not fit the 160 KB LDS.
#define DEC_A0 0                      // MB=4 KG=4     4096
#define DEC_B0 4096                   // 128
#define DEC_A1 4224                   // MB=4 KG=16    16384
#define DEC_B1 20608                  // 128
#define DEC_A2 20736                  // MB=3 KG=16    12288
#define DEC_B2 33024                  // 96
#define DEC_B3 33120                  // 128
#define DEC_HW 33248                  // sdf head   [mb 4][half 2][16]   128
#define DEC_HU 33376                  // std head                        128
#define DEC_HB 33504                  // b4, bu, 0, 0
#define DEC_LDS_FLOATS 33508
#define DEC_A3 33508                  // MB=4 KG=16    16384   (global)
#define DEC_FLOATS 49892

// One 32-point tile through the decoder.  xin[t] = x0[k = 2t + half] of point (lane&31), x0 = [latent 29 | xyz 3].
// Returns (sdf, std) for point (lane & 31), identical in both halves.
__device__ __forceinline__ void decoder_tile(const float* __restrict__ W /* LDS */, const float* __restrict__ Wg /* global blob */,
                                             const f16v& xin, int lane, float& sdf, float& stdv) {
    const int half = lane >> 5;
    f16v hx[1];
    hx[0] = xin;
    f16v h0[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        f16v acc = load_bias16(W + DEC_B0 + mb * 32, half);
        acc = block_mm<1>(reinterpret_cast<const f4v*>(W + DEC_A0) + (mb * 4) * 64, hx, acc, lane);
        h0[mb] = relu16(acc);
    }
    f16v h1[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        f16v acc = load_bias16(W + DEC_B1 + mb * 32, half);
        acc = block_mm<4>(reinterpret_cast<const f4v*>(W + DEC_A1) + (mb * 16) * 64, h0, acc, lane);
        h1[mb] = relu16(acc);
    }
    f16v h2x[4];                       // [h2 (96 features) | x0 (32)] : the latent_in=[3] skip (di_decoder.py:61-62)
#pragma unroll
    for (int mb = 0; mb < 3; ++mb) {
        f16v acc = load_bias16(W + DEC_B2 + mb * 32, half);
        acc = block_mm<4>(reinterpret_cast<const f4v*>(W + DEC_A2) + (mb * 16) * 64, h1, acc, lane);
        h2x[mb] = relu16(acc);
    }
    h2x[3] = xin;
    // L3's A operand comes from global memory and is loop-invariant per lane: without this barrier LICM hoists all 64
    // float4 loads out of the persistent tile loop and spills them (256 VGPRs) instead of streaming them through L1/L2.
    // (The base pointer is also made opaque per tile: otherwise the 64 per-load 64-bit addresses get hoisted and spilled.)
    const float* Wg3 = Wg + DEC_A3;
    asm volatile("" : "+s"(Wg3) : : "memory");
    float ps = 0.0f, pu = 0.0f;        // per-lane partial dot products of the two 128->1 heads
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        f16v acc = load_bias16(W + DEC_B3 + mb * 32, half);
        acc = block_mm<4>(reinterpret_cast<const f4v*>(Wg3) + (mb * 16) * 64, h2x, acc, lane);
        acc = relu16(acc);
        f16v ws = load_bias16(W + DEC_HW + mb * 32, half);
        f16v wu = load_bias16(W + DEC_HU + mb * 32, half);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ps = fmaf(acc[r], ws[r], ps);
            pu = fmaf(acc[r], wu[r], pu);
        }
    }
    ps += __shfl_xor(ps, 32);
    pu += __shfl_xor(pu, 32);
    ps += W[DEC_HB + 0];
    pu += W[DEC_HB + 1];
    sdf = tanhf(ps);                                                       // di_decoder.py:84
    float sp = (pu > 20.0f) ? pu : log1pf(expf(pu));                       // F.softplus (beta=1, threshold=20)
    stdv = 0.05f + 0.5f * sp;                                              // di_decoder.py:68
}

// cooperative global -> LDS copy of `n_floats` (multiple of 4) by the whole block
__device__ __forceinline__ void stage_weights(float* lds, const float* __restrict__ g, int n_floats) {
    const f4v* src = reinterpret_cast<const f4v*>(g);
    f4v* dst = reinterpret_cast<f4v*>(lds);
    for (int i = threadIdx.x; i < n_floats / 4; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}

}  // namespace dif
