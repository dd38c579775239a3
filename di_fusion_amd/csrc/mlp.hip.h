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

// max(x, 0) as ONE instruction: fmaxf costs two (IEEE mode makes the compiler canonicalise the MFMA result first, and it turns
// v_med3 back into that pair).  On the bit pattern a signed-integer max does it: every negative float (and -0) is a negative int32,
// every non-negative float is returned unchanged.  A NaN keeps its payload if its sign bit is clear (torch's relu propagates NaN too).
// Not inline asm: the hazard recogniser does not see into asm, and an asm v_max reading an accumulator right behind the MFMA that
// writes it gets stale registers (measured: encoder rows off by 0.7).
__device__ __forceinline__ float relu1(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
__device__ __forceinline__ f16v relu16(f16v v) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = relu1(v[i]);
    return v;
}

// acc += W_block (32 x 32*NB) * H_in (32*NB x 32 points).  `A` points at k-group 0 of the out-block.
// The A stream is software-pipelined by hand: the float4 for k-group t+1 is requested before the 4 MFMAs of k-group t
// (256 cycles of matrix-pipe time cover the LDS / L2 latency); sched_barrier pins that order — left alone, the
// scheduler hoists every ds_read of the fully unrolled chain to the top and spills hundreds of VGPRs.
// Where the A stream comes from: LDS (ds_read_b128) or a buffer resource over the packed blob in global memory
// (buffer_load_dwordx4 with SGPR base + scalar offset: no per-load 64-bit VGPR address, nothing for LICM to hoist and spill).
typedef unsigned int u4v __attribute__((ext_vector_type(4)));

struct LdsA {
    const f4v* p;
    __device__ __forceinline__ f4v load(int t, int lane) const { return p[t * 64 + lane]; }
};

struct BufA {
    __amdgpu_buffer_rsrc_t rsrc;
    int base;                           // byte offset of k-group 0 (wave-uniform)
    __device__ __forceinline__ f4v load(int t, int lane) const {
        u4v v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, base + t * 1024, 0);
        return __builtin_bit_cast(f4v, v);
    }
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, int n_floats) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, n_floats * 4, 0x00020000);
}

template <int NB, class ASRC>
__device__ __forceinline__ f16v block_mm_src(const ASRC& A, const f16v (&hin)[NB], f16v acc, int lane) {
    f4v a = A.load(0, lane);
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int t = kb * 4 + g;
            f4v an = a;
            if (t + 1 < NB * 4) an = A.load(t + 1, lane);
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

// All NMB out-blocks of a layer whose A operand is STREAMED from global memory (L2), as one software pipeline: the float4 of k-group
// u + PF is requested before the 4 MFMAs of k-group u.  One k-group is 256 cycles of matrix-pipe time (~0.1 us) against ~0.6 us of L2
// latency, so a kernel with a single wave per SIMD (the gradient kernels: nothing else on the SIMD hides the wait) needs PF ~ 8; the
// stream runs across the out-blocks (they are consecutive in the blob), so only the first k-groups of a layer see the full latency.
template <int NMB, int NB, int PF>
__device__ __forceinline__ void stream_mm(const BufA& A, const f16v (&hin)[NB], f16v (&acc)[NMB], int lane) {
    constexpr int NU = NMB * NB * 4;
    f4v ring[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i)
        if (i < NU) ring[i] = A.load(i, lane);
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int u = (mb * NB + kb) * 4 + g;
                const f4v a = ring[u % PF];
                if (u + PF < NU) ring[u % PF] = A.load(u + PF, lane);
                acc[mb] = mfma32(a.x, hin[kb][4 * g + 0], acc[mb]);
                acc[mb] = mfma32(a.y, hin[kb][4 * g + 1], acc[mb]);
                acc[mb] = mfma32(a.z, hin[kb][4 * g + 2], acc[mb]);
                acc[mb] = mfma32(a.w, hin[kb][4 * g + 3], acc[mb]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

template <int NB>
__device__ __forceinline__ f16v block_mm(const f4v* __restrict__ A, const f16v (&hin)[NB], f16v acc, int lane) {
    return block_mm_src<NB>(LdsA{A}, hin, acc, lane);
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
__device__ __forceinline__ void decoder_tile(const float* __restrict__ W /* LDS */, __amdgpu_buffer_rsrc_t Wg /* global blob */,
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
    // L3's A operand comes from global memory and is loop-invariant per lane: the offset is made opaque per tile, otherwise
    // LICM hoists all 64 loads out of the persistent tile loop and spills them (256 VGPRs) instead of streaming through L1/L2.
    int off3 = DEC_A3 * 4;
    asm volatile("" : "+s"(off3) : : "memory");
    float ps = 0.0f, pu = 0.0f;        // per-lane partial dot products of the two 128->1 heads
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        f16v acc = load_bias16(W + DEC_B3 + mb * 32, half);
        acc = block_mm_src<4>(BufA{Wg, off3 + mb * 16 * 1024}, h2x, acc, lane);
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

// ---- decoder with the latent folded into per-voxel constants -----------------------------------------------------------------
// Every sample row of a voxel shares the voxel's latent z, which enters the decoder twice: through lin0 (x0 = [z | xyz]) and through the
// skip connection of lin3.  c0 = b0 + W0[:, :29] z and c3 = b3 + W3[:, 96:125] z are therefore computed ONCE per voxel
// (decoder_fold_consts: 2 x 128 dot products of length 29 on the VALU, weights from packing.py:pack_decoder_fold) and become the
// accumulators' initial values; the MFMAs only add the three coordinate columns (k = 29, 30, 31: the last two k-steps of the
// natural-order block).  656 MFMAs per 32-row tile instead of 768.  The sums are associated differently from decoder_tile (latent
// terms first), so results agree to rounding (~1e-7 relative), not bit for bit.
#define DECF_FLOATS (2 * 29 * 128)

// c (256 floats, wave-private LDS): [c0 | c3] in accumulator-fragment order.  lat_row is the same for the whole wave.  B0 / B3: where
// the two biases sit in the LDS image (the f32 blob and the bf16-pipe blob differ); UNROLL loads are in flight together (fully
// unrolled they get hoisted en masse and spill).
template <int B0, int B3, int UNROLL>
__device__ __forceinline__ void decoder_fold_consts_at(const float* __restrict__ W /* LDS */, const float* __restrict__ fold /* global */,
                                                       const float* __restrict__ lat_row, float* __restrict__ c, int lane) {
    float a0 = W[B0 + lane], a1 = W[B0 + lane + 64], a2 = W[B3 + lane], a3 = W[B3 + lane + 64];
    const f4v* wk = reinterpret_cast<const f4v*>(fold) + lane;       // [k][lane] -> (lin0: p = lane, lane + 64; lin3: p = lane, lane + 64)
#pragma unroll UNROLL
    for (int k = 0; k < 29; ++k) {
        const float zk = lat_row[k];
        const f4v wv = wk[k * 64];
        a0 = fmaf(wv.x, zk, a0);
        a1 = fmaf(wv.y, zk, a1);
        a2 = fmaf(wv.z, zk, a2);
        a3 = fmaf(wv.w, zk, a3);
    }
    c[lane] = a0;
    c[lane + 64] = a1;
    c[128 + lane] = a2;
    c[128 + lane + 64] = a3;
}
__device__ __forceinline__ void decoder_fold_consts(const float* __restrict__ W, const float* __restrict__ fold, const float* __restrict__ lat_row,
                                                    float* __restrict__ c, int lane) {
    decoder_fold_consts_at<DEC_B0, DEC_B3, 8>(W, fold, lat_row, c, lane);
}

// acc init for (layer, out-block mb): from the wave's LDS record, or from a per-lane record in global memory (rows of different voxels)
struct FoldInitLds {
    const float* c;
    __device__ __forceinline__ f16v load(int layer, int mb, int half) const { return load_bias16(c + layer * 128 + mb * 32, half); }
};
struct FoldInitGlobal {
    const float* rec;               // this lane's voxel record (256 floats)
    __device__ __forceinline__ f16v load(int layer, int mb, int half) const { return load_bias16(rec + layer * 128 + mb * 32, half); }
};

// One 32-point tile; (px, py, pz) = voxel-local coordinates of point (lane & 31).
template <class INIT>
__device__ __forceinline__ void decoder_tile_folded(const float* __restrict__ W /* LDS */, __amdgpu_buffer_rsrc_t Wg /* global blob */, const INIT& init,
                                                    float px, float py, float pz, int lane, float& sdf, float& stdv) {
    const int half = lane >> 5;
    const float b14 = half ? px : 0.0f;             // k-step 14 contracts features (28, 29): the latent's last entry is folded away
    const float b15 = half ? pz : py;               // k-step 15: features (30, 31)
    f16v h0[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        f16v acc = init.load(0, mb, half);
        const f4v a = reinterpret_cast<const f4v*>(W + DEC_A0)[(mb * 4 + 3) * 64 + lane];
        acc = mfma32(a.z, b14, acc);
        acc = mfma32(a.w, b15, acc);
        h0[mb] = relu16(acc);
        __builtin_amdgcn_sched_barrier(0);
    }
    f16v h1[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        f16v acc = load_bias16(W + DEC_B1 + mb * 32, half);
        acc = block_mm<4>(reinterpret_cast<const f4v*>(W + DEC_A1) + (mb * 16) * 64, h0, acc, lane);
        h1[mb] = relu16(acc);
    }
    f16v h2[3];
#pragma unroll
    for (int mb = 0; mb < 3; ++mb) {
        f16v acc = load_bias16(W + DEC_B2 + mb * 32, half);
        acc = block_mm<4>(reinterpret_cast<const f4v*>(W + DEC_A2) + (mb * 16) * 64, h1, acc, lane);
        h2[mb] = relu16(acc);
    }
    int off3 = DEC_A3 * 4;
    asm volatile("" : "+s"(off3) : : "memory");
    float ps = 0.0f, pu = 0.0f;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        f16v acc = init.load(1, mb, half);
        const BufA src{Wg, off3 + mb * 16 * 1024};
        const f4v ax = src.load(15, lane);          // the k-group that holds the coordinate columns of the skip block
        acc = block_mm_src<3>(src, h2, acc, lane);
        acc = mfma32(ax.z, b14, acc);
        acc = mfma32(ax.w, b15, acc);
        __builtin_amdgcn_sched_barrier(0);
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

// ---- fp32 products on the bf16 matrix pipe ("x6") --------------------------------------------------------------------------------
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the f32-input MFMA.  An fp32 value is the exact sum of three bf16 slices
// (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid); the differences are exact), a product of two slices is exact in the
// fp32 accumulator, and the six slice products of weight >= 2^-16 (hi*hi, hi*mid, mid*hi, hi*lo, mid*mid, lo*hi) give the fp32
// product to ~2^-24 relative: the rounding class of the f32 MFMA (against float64, 128-term dot products: 1.05e-6 vs 1.42e-6 for
// v_mfma_f32_32x32x2_f32) in 6/16 of its pipe time.  The chaining is the same as above: registers 8s..8s+7 of a D fragment are the
// eight k values (k = 8*half + j) of k-step s of the next layer, for which packing.py:pack_A_x6 lays out the weight slices.
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
struct Tri { u4v q0, q1, q2; };                       // hi / mid / lo slices of 8 values (A: one weight row chunk, B: one point's chunk)

__device__ __forceinline__ f16v mfb(u4v a, u4v b, f16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8v, a), __builtin_bit_cast(bf8v, b), c, 0, 0, 0);
}
typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {        // v_cvt_pk_bf16_f32: RNE, lo -> bits 15:0
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{lo, hi}, bf2v));
}
// slices of the value pair (a, b) -> dword p of the three fragments
__device__ __forceinline__ void split_pair(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
    p1 = cvt_pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(p1 << 16), sb = rb - __uint_as_float(p1 & 0xffff0000u);
    p2 = cvt_pk_bf16(sa, sb);
}
__device__ __forceinline__ void split_pair_into(const f16v& h, int s, int p, Tri& t) {
    unsigned p0, p1, p2;
    split_pair(h[8 * s + 2 * p], h[8 * s + 2 * p + 1], p0, p1, p2);
    t.q0[p] = p0; t.q1[p] = p1; t.q2[p] = p2;
}

struct LdsX6 {                  // step t = 3 consecutive 1 KB fragments
    const u4v* p;
    __device__ __forceinline__ Tri load(int t, int lane) const { const u4v* q = p + t * 192 + lane; return Tri{q[0], q[64], q[128]}; }
};
struct BufX6 {
    __amdgpu_buffer_rsrc_t rsrc;
    int base;                   // byte offset of step 0 (wave-uniform)
    __device__ __forceinline__ Tri load(int t, int lane) const {
        const int o = base + t * 3072;              // one scalar offset per step; the three fragments through the instruction's immediate offset
        return Tri{__builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, o, 0), __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16 + 1024, o, 0),
                   __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16 + 2048, o, 0)};
    }
};

// the six slice products of one k-step, smallest first
__device__ __forceinline__ f16v step_x6(const Tri& a, const Tri& x, f16v c) {
    c = mfb(a.q2, x.q0, c);
    c = mfb(a.q1, x.q1, c);
    c = mfb(a.q0, x.q2, c);
    c = mfb(a.q1, x.q0, c);
    c = mfb(a.q0, x.q1, c);
    c = mfb(a.q0, x.q0, c);
    return c;
}

// acc[mo] += W[mo-block][input blocks KB0..KB1) * hin, input blocks outermost (each block is sliced once, while the previous block's
// MFMAs run: one value pair per step), weight fragments PF steps ahead of their use.  Step order = memory order of the source.
template <int KB0, int KB1, int NMO, int PF, class SRC>
__device__ __forceinline__ void layer_x6(const SRC& A, const f16v* hin, f16v* acc, int lane) {
    constexpr int NS = (KB1 - KB0) * 2 * NMO;
    Tri ring[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i)
        if (i < NS) ring[i] = A.load(i, lane);
    Tri x[2], xn[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) split_pair_into(hin[KB0], i >> 2, i & 3, x[i >> 2]);
#pragma unroll
    for (int kb = KB0; kb < KB1; ++kb) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int mo = 0; mo < NMO; ++mo) {
                const int u = s * NMO + mo;                      // step within the block, 0 .. 2*NMO-1
                const int t = (kb - KB0) * 2 * NMO + u;
                const Tri a = ring[t % PF];
                if (t + PF < NS) ring[t % PF] = A.load(t + PF, lane);
                if (kb + 1 < KB1) {                              // next block's slices, spread over this block's steps
                    constexpr int per = (8 + 2 * NMO - 1) / (2 * NMO);
#pragma unroll
                    for (int i = u * per; i < (u + 1) * per && i < 8; ++i) split_pair_into(hin[kb + 1], i >> 2, i & 3, xn[i >> 2]);
                }
                acc[mo] = step_x6(a, x[s], acc[mo]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (kb + 1 < KB1) { x[0] = xn[0]; x[1] = xn[1]; }
    }
}

// blob offsets (packing.py:pack_decoder_x6): fp32 auxiliary part in floats, slices in bytes from the blob start
#ifndef X6_PF_LDS
#define X6_PF_LDS 2
#endif
#define X6_A0C 0                      // [mb 4][lane 64] float4: k-group 3 of lin0 (the coordinate columns)
#define X6_B0 1024
#define X6_B1 1152
#define X6_B2 1280
#define X6_B3 1376
#define X6_HW 1504
#define X6_HU 1632
#define X6_HB 1760
#define X6_A3C 1764                   // [mb 4][lane 64] float4: k-group 15 of lin3
#define X6_AUX_FLOATS 2788
#define X6_L1 (X6_AUX_FLOATS * 4)     // [kb 4][s 2][mo 4][slice 3][lane 64][8 bf16]   98,304 B
#define X6_L2 (X6_L1 + 98304)         // [kb 4][s 2][mo 3]...                           73,728 B; input blocks 0,1 are staged in LDS
#define X6_LDS_BYTES (X6_L2 + 36864)  // 146,320 B
#define X6_L3 (X6_L2 + 73728)         // [kb 3][s 2][mo 4]...                           73,728 B
#define X6_BYTES (X6_L3 + 73728)      // 256,912 B

__device__ __forceinline__ void decoder_fold_consts_x6(const float* __restrict__ aux /* LDS */, const float* __restrict__ fold /* global */,
                                                       const float* __restrict__ lat_row, float* __restrict__ c, int lane) {
    decoder_fold_consts_at<X6_B0, X6_B3, 15>(aux, fold, lat_row, c, lane);       // two batches of loads (fully unrolled the kernel spills)
}

// decoder_tile_folded on the bf16 pipe.  W = LDS copy of blob[0, X6_LDS_BYTES), Wg = buffer resource over the whole blob.
template <class INIT>
__device__ __forceinline__ void decoder_tile_folded_x6(const float* __restrict__ W /* LDS */, __amdgpu_buffer_rsrc_t Wg, const INIT& init,
                                                       float px, float py, float pz, int lane, float& sdf, float& stdv) {
    const int half = lane >> 5;
    const float b14 = half ? px : 0.0f;
    const float b15 = half ? pz : py;
    const char* Wb = reinterpret_cast<const char*>(W);
    f16v h0[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        f16v acc = init.load(0, mb, half);
        const f4v a = reinterpret_cast<const f4v*>(W + X6_A0C)[mb * 64 + lane];
        acc = mfma32(a.z, b14, acc);
        acc = mfma32(a.w, b15, acc);
        h0[mb] = relu16(acc);
    }
    f16v h1[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) h1[mb] = load_bias16(W + X6_B1 + mb * 32, half);
    layer_x6<0, 4, 4, X6_PF_LDS>(LdsX6{reinterpret_cast<const u4v*>(Wb + X6_L1)}, h0, h1, lane);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) h1[mb] = relu16(h1[mb]);
    f16v h2[3];
#pragma unroll
    for (int mb = 0; mb < 3; ++mb) h2[mb] = load_bias16(W + X6_B2 + mb * 32, half);
    int goff = X6_L2 + 36864;                    // opaque per tile: see decoder_tile (LICM would hoist and spill the loop-invariant loads)
    asm volatile("" : "+s"(goff) : : "memory");
    layer_x6<0, 2, 3, X6_PF_LDS>(LdsX6{reinterpret_cast<const u4v*>(Wb + X6_L2)}, h1, h2, lane);
    layer_x6<2, 4, 3, 3>(BufX6{Wg, goff}, h1, h2, lane);
#pragma unroll
    for (int mb = 0; mb < 3; ++mb) h2[mb] = relu16(h2[mb]);
    f16v h3[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) h3[mb] = init.load(1, mb, half);
    layer_x6<0, 3, 4, 3>(BufX6{Wg, goff + 36864}, h2, h3, lane);
    float ps = 0.0f, pu = 0.0f;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        f16v acc = h3[mb];
        const f4v ax = reinterpret_cast<const f4v*>(W + X6_A3C)[mb * 64 + lane];
        acc = mfma32(ax.z, b14, acc);
        acc = mfma32(ax.w, b15, acc);
        acc = relu16(acc);
        f16v ws = load_bias16(W + X6_HW + mb * 32, half);
        f16v wu = load_bias16(W + X6_HU + mb * 32, half);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ps = fmaf(acc[r], ws[r], ps);
            pu = fmaf(acc[r], wu[r], pu);
        }
    }
    ps += __shfl_xor(ps, 32);
    pu += __shfl_xor(pu, 32);
    ps += W[X6_HB + 0];
    pu += W[X6_HB + 1];
    sdf = tanhf(ps);
    float sp = (pu > 20.0f) ? pu : log1pf(expf(pu));
    stdv = 0.05f + 0.5f * sp;
}

// The unfolded decoder tile on the bf16 pipe (explicit 32-column rows: dif_decode_rows, get_sdf values, the non-fast lattice): lin0 and
// the skip block of lin3 take the input fragment `xin` (natural k order) through slices of their own (Wu = packing.py:pack_decoder_x6u,
// streamed from L2); everything else as decoder_tile_folded_x6.
#define X6U_L0 0
#define X6U_L3X 24576
#define X6U_BYTES 49152
__device__ __forceinline__ void decoder_tile_x6(const float* __restrict__ W /* LDS */, __amdgpu_buffer_rsrc_t Wg, __amdgpu_buffer_rsrc_t Wu,
                                                const f16v& xin, int lane, float& sdf, float& stdv) {
    const int half = lane >> 5;
    const char* Wb = reinterpret_cast<const char*>(W);
    int uoff = X6U_L0, goff = X6_L2 + 36864;     // opaque per tile (see decoder_tile)
    asm volatile("" : "+s"(uoff), "+s"(goff) : : "memory");
    f16v hx[1];
    hx[0] = xin;
    f16v h0[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) h0[mb] = load_bias16(W + X6_B0 + mb * 32, half);
    layer_x6<0, 1, 4, 3>(BufX6{Wu, uoff}, hx, h0, lane);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) h0[mb] = relu16(h0[mb]);
    f16v h1[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) h1[mb] = load_bias16(W + X6_B1 + mb * 32, half);
    layer_x6<0, 4, 4, X6_PF_LDS>(LdsX6{reinterpret_cast<const u4v*>(Wb + X6_L1)}, h0, h1, lane);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) h1[mb] = relu16(h1[mb]);
    f16v h2[3];
#pragma unroll
    for (int mb = 0; mb < 3; ++mb) h2[mb] = load_bias16(W + X6_B2 + mb * 32, half);
    layer_x6<0, 2, 3, X6_PF_LDS>(LdsX6{reinterpret_cast<const u4v*>(Wb + X6_L2)}, h1, h2, lane);
    layer_x6<2, 4, 3, 3>(BufX6{Wg, goff}, h1, h2, lane);
#pragma unroll
    for (int mb = 0; mb < 3; ++mb) h2[mb] = relu16(h2[mb]);
    f16v h3[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) h3[mb] = load_bias16(W + X6_B3 + mb * 32, half);
    layer_x6<0, 3, 4, 3>(BufX6{Wg, goff + 36864}, h2, h3, lane);
    layer_x6<0, 1, 4, 3>(BufX6{Wu, uoff + X6U_L3X}, hx, h3, lane);
    float ps = 0.0f, pu = 0.0f;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const f16v acc = relu16(h3[mb]);
        f16v ws = load_bias16(W + X6_HW + mb * 32, half);
        f16v wu = load_bias16(W + X6_HU + mb * 32, half);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ps = fmaf(acc[r], ws[r], ps);
            pu = fmaf(acc[r], wu[r], pu);
        }
    }
    ps += __shfl_xor(ps, 32);
    pu += __shfl_xor(pu, 32);
    ps += W[X6_HB + 0];
    pu += W[X6_HB + 1];
    sdf = tanhf(ps);
    float sp = (pu > 20.0f) ? pu : log1pf(expf(pu));
    stdv = 0.05f + 0.5f * sp;
}

// Encoder on the bf16 pipe (blob = packing.py:pack_encoder_x6, all of it staged in LDS).  lin0 (6 -> 32, three k-steps) stays on the f32
// MFMA.  lin2's out-block mb+1 is computed while out-block mb is being sliced for lin3, so the weight steps are stored in the order
// they are consumed: lin1 | L2(0) | L2(1) L3(0) | L2(2) L3(1) | ... | L2(7) L3(6) | L3(7)   (L2(mb): 4 steps, L3(mb): 2 steps).
#define E6_A0 0
#define E6_B0 256
#define E6_B1 288
#define E6_B2 352
#define E6_B3 608
#define E6_AUX_FLOATS 640
#define E6_L1 (E6_AUX_FLOATS * 4)     // [s 2][mo 2][slice 3][lane 64][8 bf16]    12,288 B
#define E6_L23 (E6_L1 + 12288)        // 48 steps of 3 KB                         147,456 B
#define E6_BYTES (E6_L23 + 147456)    // 162,304 B

__device__ __forceinline__ f16v encoder_tile_x6(const float* __restrict__ W /* LDS */, float x0, float x1, float x2, int lane) {
    const int half = lane >> 5;
    const char* Wb = reinterpret_cast<const char*>(W);
    f16v h0[1];
    {
        f16v acc = load_bias16(W + E6_B0, half);
        f4v a = reinterpret_cast<const f4v*>(W + E6_A0)[lane];
        acc = mfma32(a.x, x0, acc);
        acc = mfma32(a.y, x1, acc);
        acc = mfma32(a.z, x2, acc);
        h0[0] = relu16(acc);
    }
    f16v h1[2];
    h1[0] = load_bias16(W + E6_B1, half);
    h1[1] = load_bias16(W + E6_B1 + 32, half);
    layer_x6<0, 1, 2, 1>(LdsX6{reinterpret_cast<const u4v*>(Wb + E6_L1)}, h0, h1, lane);
    h1[0] = relu16(h1[0]);
    h1[1] = relu16(h1[1]);
    Tri xs[4];                                  // h1 sliced once: [kb][s]
#pragma unroll
    for (int i = 0; i < 16; ++i) split_pair_into(h1[i >> 3], (i >> 2) & 1, i & 3, xs[i >> 2]);
    const LdsX6 A{reinterpret_cast<const u4v*>(Wb + E6_L23)};
    int t = 0;
    Tri a = A.load(0, lane), an;
    f16v cur = load_bias16(W + E6_B2, half);
#pragma unroll
    for (int u = 0; u < 4; ++u) {               // L2(0)
        an = A.load(t + 1, lane);
        cur = step_x6(a, xs[u], cur);
        a = an; ++t;
        __builtin_amdgcn_sched_barrier(0);
    }
    f16v out = load_bias16(W + E6_B3, half);
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
        const f16v h = relu16(cur);
        Tri hs[2];
        if (mb + 1 < 8) {
            cur = load_bias16(W + E6_B2 + (mb + 1) * 32, half);
#pragma unroll
            for (int u = 0; u < 4; ++u) {       // L2(mb + 1), with the slicing of h2 block mb spread over its steps
                an = A.load(t + 1, lane);
                split_pair_into(h, u >> 1, (2 * u) & 3, hs[u >> 1]);
                split_pair_into(h, u >> 1, (2 * u + 1) & 3, hs[u >> 1]);
                cur = step_x6(a, xs[u], cur);
                a = an; ++t;
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) split_pair_into(h, i >> 2, i & 3, hs[i >> 2]);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {           // L3(mb)
            an = a;
            if (t + 1 < 48) an = A.load(t + 1, lane);
            out = step_x6(a, hs[s], out);
            a = an; ++t;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    return out;
}

// ---- decoder with input gradient (get_sdf for the tracker: d sdf / d xyz, reference tracker.py:186-192) -------------------
// Backward blob (global memory, packing.py:pack_decoder_backward): transposed layers, k order = D-fragment order of the
// forward layer's OUTPUT blocks, so the masked upstream gradient fragments are again ready-made B operands.
#define DECB_T3 0                     // W3^T  MB=4 (rows: h2 0..95 | x0 96..127)  KG=16   16384
#define DECB_T2 16384                 // W2^T  MB=4 (h1)   KG=12                           12288
#define DECB_T1 28672                 // W1^T  MB=4 (h0)   KG=16                           16384
#define DECB_T0 45056                 // W0^T  MB=1 (x0)   KG=16                            4096
#define DECB_FLOATS 49152

__device__ __forceinline__ f16v relu16_mask(f16v v, unsigned& mask) {
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        m |= (v[i] > 0.0f) ? (1u << i) : 0u;
        v[i] = fmaxf(v[i], 0.0f);
    }
    mask = m;
    return v;
}

__device__ __forceinline__ f16v apply_mask16(f16v v, unsigned mask) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = ((mask >> i) & 1u) ? v[i] : 0.0f;
    return v;
}

__device__ __forceinline__ f16v zero16() {
    f16v z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.0f;
    return z;
}

#define GRAD_PF 8
// Forward as decoder_tile (recording the ReLU masks, 16 bits per out-block), then the reverse chain
//   g3 = w4 (.) mask3 ; [g2 | gx_skip] = W3^T g3 ; g1 = W2^T (g2 (.) mask2) ; g0 = W1^T (g1 (.) mask1) ; gx = W0^T (g0 (.) mask0)
// d sdf / d x0[29..31] = (1 - sdf^2) * (gx + gx_skip)[29..31]   (tanh').  Features 29,30,31 of a natural-order block sit in
// registers 13,14,15 of the lanes 32..63.  Returns them in (gx, gy, gz) on those lanes (garbage on lanes 0..31).
__device__ __forceinline__ void decoder_tile_grad(const float* __restrict__ W /* LDS */, __amdgpu_buffer_rsrc_t Wg /* fwd blob */,
                                                  __amdgpu_buffer_rsrc_t Wb /* bwd blob */, const f16v& xin, int lane,
                                                  float& sdf, float& stdv, float& gx, float& gy, float& gz) {
    const int half = lane >> 5;
    unsigned m0[4], m1[4], m2[3], m3[4];
    f16v hx[1];
    hx[0] = xin;
    f16v h0[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        f16v acc = load_bias16(W + DEC_B0 + mb * 32, half);
        acc = block_mm<1>(reinterpret_cast<const f4v*>(W + DEC_A0) + (mb * 4) * 64, hx, acc, lane);
        h0[mb] = relu16_mask(acc, m0[mb]);
    }
    f16v h1[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        f16v acc = load_bias16(W + DEC_B1 + mb * 32, half);
        acc = block_mm<4>(reinterpret_cast<const f4v*>(W + DEC_A1) + (mb * 16) * 64, h0, acc, lane);
        h1[mb] = relu16_mask(acc, m1[mb]);
    }
    f16v h2x[4];
#pragma unroll
    for (int mb = 0; mb < 3; ++mb) {
        f16v acc = load_bias16(W + DEC_B2 + mb * 32, half);
        acc = block_mm<4>(reinterpret_cast<const f4v*>(W + DEC_A2) + (mb * 16) * 64, h1, acc, lane);
        h2x[mb] = relu16_mask(acc, m2[mb]);
    }
    h2x[3] = xin;
    int off3 = DEC_A3 * 4, offb = 0;
    asm volatile("" : "+s"(off3), "+s"(offb) : : "memory");
    float ps = 0.0f, pu = 0.0f;
    f16v g3[4];
    f16v h3[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) h3[mb] = load_bias16(W + DEC_B3 + mb * 32, half);
    stream_mm<4, 4, GRAD_PF>(BufA{Wg, off3}, h2x, h3, lane);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        f16v acc = relu16_mask(h3[mb], m3[mb]);
        f16v ws = load_bias16(W + DEC_HW + mb * 32, half);
        f16v wu = load_bias16(W + DEC_HU + mb * 32, half);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ps = fmaf(acc[r], ws[r], ps);
            pu = fmaf(acc[r], wu[r], pu);
        }
        g3[mb] = apply_mask16(ws, m3[mb]);                 // d sdf_pre / d (pre-activation of lin3)
    }
    ps += __shfl_xor(ps, 32);
    pu += __shfl_xor(pu, 32);
    ps += W[DEC_HB + 0];
    pu += W[DEC_HB + 1];
    sdf = tanhf(ps);
    float sp = (pu > 20.0f) ? pu : log1pf(expf(pu));
    stdv = 0.05f + 0.5f * sp;
    // ---- reverse chain ----
    f16v t3[4];                         // W3^T g3: rows h2 (3 blocks) | x0 (the skip block)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) t3[mb] = zero16();
    stream_mm<4, 4, GRAD_PF>(BufA{Wb, offb + DECB_T3 * 4}, g3, t3, lane);
    f16v g2[3];
#pragma unroll
    for (int mb = 0; mb < 3; ++mb) g2[mb] = apply_mask16(t3[mb], m2[mb]);
    f16v g1[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) g1[mb] = zero16();
    stream_mm<4, 3, GRAD_PF>(BufA{Wb, offb + DECB_T2 * 4}, g2, g1, lane);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) g1[mb] = apply_mask16(g1[mb], m1[mb]);
    f16v g0[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) g0[mb] = zero16();
    stream_mm<4, 4, GRAD_PF>(BufA{Wb, offb + DECB_T1 * 4}, g1, g0, lane);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) g0[mb] = apply_mask16(g0[mb], m0[mb]);
    f16v gx1[1];
    gx1[0] = t3[3];
    stream_mm<1, 4, GRAD_PF>(BufA{Wb, offb + DECB_T0 * 4}, g0, gx1, lane);
    const f16v gxv = gx1[0];
    const float dt = 1.0f - sdf * sdf;
    gx = dt * gxv[13];
    gy = dt * gxv[14];
    gz = dt * gxv[15];
}

// The same on the bf16 matrix pipe (forward as decoder_tile_x6 with the masks recorded, reverse chain over the sliced transposed
// layers of packing.py:pack_decoder_x6_backward, streamed from L2).  PF: weight steps in flight — a kernel with one wave per SIMD has
// nothing else to hide the L2 latency behind.
#define X6B_T3 0                      // [kb 4][s 2][mo 4] steps of 3 KB     98,304 B
#define X6B_T2 98304                  // [kb 3][s 2][mo 4]                   73,728 B
#define X6B_T1 172032                 // [kb 4][s 2][mo 4]                   98,304 B
#define X6B_T0 270336                 // [kb 4][s 2][mo 1]                   24,576 B
#define X6B_BYTES 294912
template <int PF>
__device__ __forceinline__ void decoder_tile_grad_x6(const float* __restrict__ W /* LDS */, __amdgpu_buffer_rsrc_t Wg, __amdgpu_buffer_rsrc_t Wu,
                                                     __amdgpu_buffer_rsrc_t Wb, const f16v& xin, int lane,
                                                     float& sdf, float& stdv, float& gx, float& gy, float& gz) {
    const int half = lane >> 5;
    const char* Wc = reinterpret_cast<const char*>(W);
    int uoff = X6U_L0, goff = X6_L2 + 36864, boff = 0;       // opaque per tile (see decoder_tile)
    asm volatile("" : "+s"(uoff), "+s"(goff), "+s"(boff) : : "memory");
    unsigned m0[4], m1[4], m2[3], m3[4];
    f16v hx[1];
    hx[0] = xin;
    f16v h0[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) h0[mb] = load_bias16(W + X6_B0 + mb * 32, half);
    layer_x6<0, 1, 4, PF>(BufX6{Wu, uoff}, hx, h0, lane);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) h0[mb] = relu16_mask(h0[mb], m0[mb]);
    f16v h1[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) h1[mb] = load_bias16(W + X6_B1 + mb * 32, half);
    layer_x6<0, 4, 4, X6_PF_LDS>(LdsX6{reinterpret_cast<const u4v*>(Wc + X6_L1)}, h0, h1, lane);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) h1[mb] = relu16_mask(h1[mb], m1[mb]);
    f16v h2[3];
#pragma unroll
    for (int mb = 0; mb < 3; ++mb) h2[mb] = load_bias16(W + X6_B2 + mb * 32, half);
    layer_x6<0, 2, 3, X6_PF_LDS>(LdsX6{reinterpret_cast<const u4v*>(Wc + X6_L2)}, h1, h2, lane);
    layer_x6<2, 4, 3, PF>(BufX6{Wg, goff}, h1, h2, lane);
#pragma unroll
    for (int mb = 0; mb < 3; ++mb) h2[mb] = relu16_mask(h2[mb], m2[mb]);
    f16v h3[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) h3[mb] = load_bias16(W + X6_B3 + mb * 32, half);
    layer_x6<0, 3, 4, PF>(BufX6{Wg, goff + 36864}, h2, h3, lane);
    layer_x6<0, 1, 4, PF>(BufX6{Wu, uoff + X6U_L3X}, hx, h3, lane);
    float ps = 0.0f, pu = 0.0f;
    f16v g3[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const f16v acc = relu16_mask(h3[mb], m3[mb]);
        const f16v ws = load_bias16(W + X6_HW + mb * 32, half);
        const f16v wu = load_bias16(W + X6_HU + mb * 32, half);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ps = fmaf(acc[r], ws[r], ps);
            pu = fmaf(acc[r], wu[r], pu);
        }
        g3[mb] = apply_mask16(ws, m3[mb]);                 // d sdf_pre / d (pre-activation of lin3)
    }
    ps += __shfl_xor(ps, 32);
    pu += __shfl_xor(pu, 32);
    ps += W[X6_HB + 0];
    pu += W[X6_HB + 1];
    sdf = tanhf(ps);
    const float sp = (pu > 20.0f) ? pu : log1pf(expf(pu));
    stdv = 0.05f + 0.5f * sp;
    // ---- reverse chain ----
    f16v t3[4];                         // W3^T g3: rows h2 (3 blocks) | x0 (the skip block)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) t3[mb] = zero16();
    layer_x6<0, 4, 4, PF>(BufX6{Wb, boff + X6B_T3}, g3, t3, lane);
    f16v g2[3];
#pragma unroll
    for (int mb = 0; mb < 3; ++mb) g2[mb] = apply_mask16(t3[mb], m2[mb]);
    f16v g1[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) g1[mb] = zero16();
    layer_x6<0, 3, 4, PF>(BufX6{Wb, boff + X6B_T2}, g2, g1, lane);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) g1[mb] = apply_mask16(g1[mb], m1[mb]);
    f16v g0[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) g0[mb] = zero16();
    layer_x6<0, 4, 4, PF>(BufX6{Wb, boff + X6B_T1}, g1, g0, lane);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) g0[mb] = apply_mask16(g0[mb], m0[mb]);
    f16v gx1[1];
    gx1[0] = t3[3];
    layer_x6<0, 4, 1, PF>(BufX6{Wb, boff + X6B_T0}, g0, gx1, lane);
    const float dt = 1.0f - sdf * sdf;
    gx = dt * gx1[0][13];
    gy = dt * gx1[0][14];
    gz = dt * gx1[0][15];
}

// The optimiser's tile (decoder_tile_nll_grad below) on the bf16 matrix pipe: forward and reverse chain as decoder_tile_grad_x6, the
// upstream gradient a * w_sdf + b * w_std formed after both heads are known.  Returns d loss / d x0 as a D fragment.
template <int PF>
__device__ __forceinline__ void decoder_tile_nll_grad_x6(const float* __restrict__ W /* LDS */, __amdgpu_buffer_rsrc_t Wg, __amdgpu_buffer_rsrc_t Wu,
                                                         __amdgpu_buffer_rsrc_t Wb, const f16v& xin, int lane, float gt, float inv_n,
                                                         float& sdf, float& stdv, float& loss, f16v& gx) {
    const int half = lane >> 5;
    const char* Wc = reinterpret_cast<const char*>(W);
    int uoff = X6U_L0, goff = X6_L2 + 36864, boff = 0;       // opaque per tile (see decoder_tile)
    asm volatile("" : "+s"(uoff), "+s"(goff), "+s"(boff) : : "memory");
    unsigned m0[4], m1[4], m2[3], m3[4];
    f16v hx[1];
    hx[0] = xin;
    f16v h0[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) h0[mb] = load_bias16(W + X6_B0 + mb * 32, half);
    layer_x6<0, 1, 4, PF>(BufX6{Wu, uoff}, hx, h0, lane);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) h0[mb] = relu16_mask(h0[mb], m0[mb]);
    f16v h1[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) h1[mb] = load_bias16(W + X6_B1 + mb * 32, half);
    layer_x6<0, 4, 4, X6_PF_LDS>(LdsX6{reinterpret_cast<const u4v*>(Wc + X6_L1)}, h0, h1, lane);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) h1[mb] = relu16_mask(h1[mb], m1[mb]);
    f16v h2[3];
#pragma unroll
    for (int mb = 0; mb < 3; ++mb) h2[mb] = load_bias16(W + X6_B2 + mb * 32, half);
    layer_x6<0, 2, 3, X6_PF_LDS>(LdsX6{reinterpret_cast<const u4v*>(Wc + X6_L2)}, h1, h2, lane);
    layer_x6<2, 4, 3, PF>(BufX6{Wg, goff}, h1, h2, lane);
#pragma unroll
    for (int mb = 0; mb < 3; ++mb) h2[mb] = relu16_mask(h2[mb], m2[mb]);
    f16v h3[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) h3[mb] = load_bias16(W + X6_B3 + mb * 32, half);
    layer_x6<0, 3, 4, PF>(BufX6{Wg, goff + 36864}, h2, h3, lane);
    layer_x6<0, 1, 4, PF>(BufX6{Wu, uoff + X6U_L3X}, hx, h3, lane);
    float ps = 0.0f, pu = 0.0f;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const f16v acc = relu16_mask(h3[mb], m3[mb]);
        const f16v ws = load_bias16(W + X6_HW + mb * 32, half);
        const f16v wu = load_bias16(W + X6_HU + mb * 32, half);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ps = fmaf(acc[r], ws[r], ps);
            pu = fmaf(acc[r], wu[r], pu);
        }
    }
    ps += __shfl_xor(ps, 32);
    pu += __shfl_xor(pu, 32);
    ps += W[X6_HB + 0];
    pu += W[X6_HB + 1];
    sdf = tanhf(ps);
    const float sp = (pu > 20.0f) ? pu : log1pf(expf(pu));
    stdv = 0.05f + 0.5f * sp;
    // ---- loss and its derivative w.r.t. the two pre-activations (as decoder_tile_nll_grad) ----
    const float g = fminf(fmaxf(gt, -0.2f), 0.2f), mu = fminf(fmaxf(sdf, -0.2f), 0.2f);
    const float diff = g - mu, var = stdv * stdv;
    loss = (logf(stdv) + 0.918938533204672742f + diff * diff / (2.0f * var)) * inv_n;          // 0.5 * log(2 pi)
    const float d_mu = (sdf >= -0.2f && sdf <= 0.2f) ? -diff / var : 0.0f;
    const float d_sigma = 1.0f / stdv - diff * diff / (var * stdv);
    const float a = inv_n * d_mu * (1.0f - sdf * sdf);
    const float b = inv_n * d_sigma * 0.5f * ((pu > 20.0f) ? 1.0f : 1.0f / (1.0f + expf(-pu)));
    // ---- reverse chain ----
    f16v g3[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const f16v ws = load_bias16(W + X6_HW + mb * 32, half);
        const f16v wu = load_bias16(W + X6_HU + mb * 32, half);
        f16v t;
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = a * ws[r] + b * wu[r];
        g3[mb] = apply_mask16(t, m3[mb]);
    }
    f16v t3[4];                         // W3^T g3: rows h2 (3 blocks) | x0 (the skip block)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) t3[mb] = zero16();
    layer_x6<0, 4, 4, PF>(BufX6{Wb, boff + X6B_T3}, g3, t3, lane);
    f16v g2[3];
#pragma unroll
    for (int mb = 0; mb < 3; ++mb) g2[mb] = apply_mask16(t3[mb], m2[mb]);
    f16v g1[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) g1[mb] = zero16();
    layer_x6<0, 3, 4, PF>(BufX6{Wb, boff + X6B_T2}, g2, g1, lane);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) g1[mb] = apply_mask16(g1[mb], m1[mb]);
    f16v g0[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) g0[mb] = zero16();
    layer_x6<0, 4, 4, PF>(BufX6{Wb, boff + X6B_T1}, g1, g0, lane);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) g0[mb] = apply_mask16(g0[mb], m0[mb]);
    f16v gx1[1];
    gx1[0] = t3[3];
    layer_x6<0, 4, 1, PF>(BufX6{Wb, boff + X6B_T0}, g0, gx1, lane);
    gx = gx1[0];
}

// ---- decoder with the gradient of the optimiser's loss w.r.t. ALL 32 inputs (latent optimisation, reference map.py:80-113) -------------
// Loss per row (map.py:87-96): -log N(clamp(gt, +-0.2); clamp(sdf, +-0.2), std) * inv_n.  Forward as decoder_tile_grad; the upstream
// gradient entering lin3's output is  a * w_sdf + b * w_std  with  a = dL/d(sdf pre-activation), b = dL/d(std pre-activation), both
// known only after the two heads — so the head weights are re-read for the reverse chain.  Returns d loss / d x0 as a D fragment
// (register r of lane l = feature (r&3) + 8*(r>>2) + 4*(l>>5) of point l&31) and the row's loss term.
__device__ __forceinline__ void decoder_tile_nll_grad(const float* __restrict__ W /* LDS */, __amdgpu_buffer_rsrc_t Wg /* fwd blob */,
                                                      __amdgpu_buffer_rsrc_t Wb /* bwd blob */, const f16v& xin, int lane, float gt, float inv_n,
                                                      float& sdf, float& stdv, float& loss, f16v& gx) {
    const int half = lane >> 5;
    unsigned m0[4], m1[4], m2[3], m3[4];
    f16v hx[1];
    hx[0] = xin;
    f16v h0[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        f16v acc = load_bias16(W + DEC_B0 + mb * 32, half);
        acc = block_mm<1>(reinterpret_cast<const f4v*>(W + DEC_A0) + (mb * 4) * 64, hx, acc, lane);
        h0[mb] = relu16_mask(acc, m0[mb]);
    }
    f16v h1[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        f16v acc = load_bias16(W + DEC_B1 + mb * 32, half);
        acc = block_mm<4>(reinterpret_cast<const f4v*>(W + DEC_A1) + (mb * 16) * 64, h0, acc, lane);
        h1[mb] = relu16_mask(acc, m1[mb]);
    }
    f16v h2x[4];
#pragma unroll
    for (int mb = 0; mb < 3; ++mb) {
        f16v acc = load_bias16(W + DEC_B2 + mb * 32, half);
        acc = block_mm<4>(reinterpret_cast<const f4v*>(W + DEC_A2) + (mb * 16) * 64, h1, acc, lane);
        h2x[mb] = relu16_mask(acc, m2[mb]);
    }
    h2x[3] = xin;
    int off3 = DEC_A3 * 4, offb = 0;
    asm volatile("" : "+s"(off3), "+s"(offb) : : "memory");
    float ps = 0.0f, pu = 0.0f;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        f16v acc = load_bias16(W + DEC_B3 + mb * 32, half);
        acc = block_mm_src<4>(BufA{Wg, off3 + mb * 16 * 1024}, h2x, acc, lane);
        acc = relu16_mask(acc, m3[mb]);
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
    sdf = tanhf(ps);
    const float sp = (pu > 20.0f) ? pu : log1pf(expf(pu));
    stdv = 0.05f + 0.5f * sp;
    // ---- loss and its derivative w.r.t. the two pre-activations ----
    const float g = fminf(fmaxf(gt, -0.2f), 0.2f), mu = fminf(fmaxf(sdf, -0.2f), 0.2f);
    const float diff = g - mu, var = stdv * stdv;
    loss = (logf(stdv) + 0.918938533204672742f + diff * diff / (2.0f * var)) * inv_n;          // 0.5 * log(2 pi)
    const float d_mu = (sdf >= -0.2f && sdf <= 0.2f) ? -diff / var : 0.0f;                      // clamp passes the gradient inside the interval
    const float d_sigma = 1.0f / stdv - diff * diff / (var * stdv);
    const float a = inv_n * d_mu * (1.0f - sdf * sdf);                                          // tanh'
    const float b = inv_n * d_sigma * 0.5f * ((pu > 20.0f) ? 1.0f : 1.0f / (1.0f + expf(-pu)));   // softplus'
    // ---- reverse chain ----
    f16v g3[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        f16v ws = load_bias16(W + DEC_HW + mb * 32, half);
        f16v wu = load_bias16(W + DEC_HU + mb * 32, half);
        f16v t;
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = a * ws[r] + b * wu[r];
        g3[mb] = apply_mask16(t, m3[mb]);
    }
    f16v g2[3];
#pragma unroll
    for (int mb = 0; mb < 3; ++mb)
        g2[mb] = apply_mask16(block_mm_src<4>(BufA{Wb, offb + DECB_T3 * 4 + mb * 16 * 1024}, g3, zero16(), lane), m2[mb]);
    f16v gskip = block_mm_src<4>(BufA{Wb, offb + DECB_T3 * 4 + 3 * 16 * 1024}, g3, zero16(), lane);
    f16v g1[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
        g1[mb] = apply_mask16(block_mm_src<3>(BufA{Wb, offb + DECB_T2 * 4 + mb * 12 * 1024}, g2, zero16(), lane), m1[mb]);
    f16v g0[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
        g0[mb] = apply_mask16(block_mm_src<4>(BufA{Wb, offb + DECB_T1 * 4 + mb * 16 * 1024}, g1, zero16(), lane), m0[mb]);
    gx = block_mm_src<4>(BufA{Wb, offb + DECB_T0 * 4}, g0, gskip, lane);
}

// cooperative global -> LDS copy of `n_floats` (multiple of 4) by the whole block.  Eight 16-byte loads are in flight per thread before
// the first LDS write: the copy is latency-bound (134 KB per CU is ~1 us of L2 bandwidth), so batching the loads is what shortens it.
__device__ __forceinline__ void stage_weights(float* lds, const float* __restrict__ g, int n_floats) {
    const f4v* __restrict__ src = reinterpret_cast<const f4v*>(g);
    f4v* dst = reinterpret_cast<f4v*>(lds);
    const int n4 = n_floats / 4, step = (int)blockDim.x;
    int i = (int)threadIdx.x;
    for (; i + 7 * step < n4; i += 8 * step) {
        f4v v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = src[i + k * step];
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[i + k * step] = v[k];
    }
    for (; i < n4; i += step) dst[i] = src[i];
    __syncthreads();
}

}  // namespace dif
