"""Fold and pack the DI-Fusion encoder/decoder weights for the MFMA kernels in `csrc/mlp.hip.h`.

Folding (done once at load, float64 then cast to float32):
  * decoder linears are `weight_norm`ed (reference `network/di_decoder.py:37-40`): W = g * v / ||v||_row;
  * encoder 1x1 convs are followed by eval-mode BatchNorm (reference `utils/pt_util.py:76-127,193-206`):
    W' = W * s, b' = beta - mean * s,  s = gamma / sqrt(var + 1e-5).

Packing ("transposed chaining", see mlp.hip.h): each layer's weight matrix is the MFMA A operand.  For out-block `mb`
(32 output features), k-group `g` (4 k-steps of the 32x32x2 MFMA) lane `l` stores a float4 whose component j is
    W[mb*32 + (l & 31)][kmap(4g + j, l >> 5)]
with
    natural input (read from memory):   kmap(t, half) = 2t + half
    D-fragment input (previous layer):  kmap(t, half) = 32*(t // 16) + (t%16 & 3) + 8*((t%16) >> 2) + 4*half
so that register r of the previous layer's accumulator IS the B operand of k-step r.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

ENC_FLOATS = 27264
DEC_LDS_FLOATS = 33508
DEC_FLOATS = 49892
DECB_FLOATS = 49152
DECF_FLOATS = 2 * 29 * 128
# split-bf16 decoder blob (mlp.hip.h, "x6"): fp32 auxiliary part, then the bf16 slices of lin1 | lin2 | lin3
X6_AUX_FLOATS = 2788
X6_L1_BYTES, X6_L2_BYTES, X6_L3_BYTES = 4 * 2 * 4 * 3 * 1024, 4 * 2 * 3 * 3 * 1024, 3 * 2 * 4 * 3 * 1024
X6_BYTES = X6_AUX_FLOATS * 4 + X6_L1_BYTES + X6_L2_BYTES + X6_L3_BYTES
X6U_BYTES = 2 * 2 * 4 * 3 * 1024       # lin0 | lin3's skip block, natural input order (the unfolded tile: explicit rows, get_sdf)
X6B_BYTES = (4 * 2 * 4 + 3 * 2 * 4 + 4 * 2 * 4 + 4 * 2 * 1) * 3 * 1024      # transposed layers for the input-gradient chain on the bf16 pipe
E6_AUX_FLOATS = 640
E6_BYTES = E6_AUX_FLOATS * 4 + 12288 + 147456


def _frag_feature(r: int, half: int) -> int:
    return (r & 3) + 8 * (r >> 2) + 4 * half


def kmap_natural(t: int, half: int) -> int:
    return 2 * t + half


def kmap_dfrag(t: int, half: int) -> int:
    return 32 * (t // 16) + _frag_feature(t % 16, half)


def pack_A(W: np.ndarray, MB: int, KG: int, kmap) -> np.ndarray:
    """W (M_out, K_in) -> (MB, KG, 64, 4) float32."""
    M, K = W.shape
    out = np.zeros((MB, KG, 64, 4), dtype=np.float32)
    for mb in range(MB):
        for g in range(KG):
            for lane in range(64):
                m = mb * 32 + (lane & 31)
                if m >= M:
                    continue
                for j in range(4):
                    k = kmap(4 * g + j, lane >> 5)
                    if k is not None and 0 <= k < K:
                        out[mb, g, lane, j] = W[m, k]
    return out


def pack_vec(b: np.ndarray, MB: int) -> np.ndarray:
    """b (M_out,) -> (MB, 2, 16): the accumulator-fragment order of a per-feature vector."""
    out = np.zeros((MB, 2, 16), dtype=np.float32)
    for mb in range(MB):
        for half in range(2):
            for r in range(16):
                f = mb * 32 + _frag_feature(r, half)
                if f < b.shape[0]:
                    out[mb, half, r] = b[f]
    return out


def fold_decoder(w: Dict[str, np.ndarray]):
    Ws, bs = [], []
    for i in range(5):
        if f"decoder.lin{i}.weight_v" in w:           # nn.utils.weight_norm (di_decoder.py:37-40): W = g * v / |v|_row
            v = w[f"decoder.lin{i}.weight_v"].astype(np.float64)
            g = w[f"decoder.lin{i}.weight_g"].astype(np.float64)
            Ws.append((v * (g / np.linalg.norm(v, axis=1, keepdims=True))).astype(np.float32))
        else:                                         # "weight_norm": false in network_specs
            Ws.append(w[f"decoder.lin{i}.weight"].astype(np.float32))
        bs.append(w[f"decoder.lin{i}.bias"].astype(np.float32))
    return Ws, bs, w["decoder.uncertainty_layer.weight"].astype(np.float32), w["decoder.uncertainty_layer.bias"].astype(np.float32)


def fold_encoder(w: Dict[str, np.ndarray]):
    Ws, bs = [], []
    for i in range(3):
        W = w[f"encoder.mlp.layer{i}.conv.weight"][:, :, 0].astype(np.float64)
        gamma = w[f"encoder.mlp.layer{i}.normlayer.bn.weight"].astype(np.float64)
        beta = w[f"encoder.mlp.layer{i}.normlayer.bn.bias"].astype(np.float64)
        mean = w[f"encoder.mlp.layer{i}.normlayer.bn.running_mean"].astype(np.float64)
        var = w[f"encoder.mlp.layer{i}.normlayer.bn.running_var"].astype(np.float64)
        s = gamma / np.sqrt(var + 1e-5)
        Ws.append((W * s[:, None]).astype(np.float32))
        bs.append((beta - mean * s).astype(np.float32))
    Ws.append(w["encoder.mlp.layer3.conv.weight"][:, :, 0].astype(np.float32))
    bs.append(w["encoder.mlp.layer3.conv.bias"].astype(np.float32))
    return Ws, bs


def pack_encoder(w: Dict[str, np.ndarray]) -> np.ndarray:
    Ws, bs = fold_encoder(w)
    assert [x.shape for x in Ws] == [(32, 6), (64, 32), (256, 64), (29, 256)]
    parts = [pack_A(Ws[0], 1, 1, kmap_natural), pack_vec(bs[0], 1),
             pack_A(Ws[1], 2, 4, kmap_dfrag), pack_vec(bs[1], 2),
             pack_A(Ws[2], 8, 8, kmap_dfrag), pack_vec(bs[2], 8),
             pack_A(Ws[3], 1, 32, kmap_dfrag), pack_vec(bs[3], 1)]
    blob = np.concatenate([p.reshape(-1) for p in parts]).astype(np.float32)
    assert blob.shape[0] == ENC_FLOATS, blob.shape
    return blob


def pack_decoder(w: Dict[str, np.ndarray]) -> np.ndarray:
    Ws, bs, Wu, bu = fold_decoder(w)
    assert [x.shape for x in Ws] == [(128, 32), (128, 128), (96, 128), (128, 128), (1, 128)]

    def kmap_l3(t, half):       # input of lin3 = cat([h2 (96), x0 (32)])  (di_decoder.py:61-62)
        return kmap_dfrag(t, half) if t < 48 else 96 + kmap_natural(t - 48, half)

    parts = [pack_A(Ws[0], 4, 4, kmap_natural), pack_vec(bs[0], 4),
             pack_A(Ws[1], 4, 16, kmap_dfrag), pack_vec(bs[1], 4),
             pack_A(Ws[2], 3, 16, kmap_dfrag), pack_vec(bs[2], 3),
             pack_vec(bs[3], 4),
             pack_vec(Ws[4][0], 4), pack_vec(Wu[0], 4),
             np.array([bs[4][0], bu[0], 0.0, 0.0], dtype=np.float32)]
    lds = np.concatenate([p.reshape(-1) for p in parts]).astype(np.float32)
    assert lds.shape[0] == DEC_LDS_FLOATS, lds.shape
    blob = np.concatenate([lds, pack_A(Ws[3], 4, 16, kmap_l3).reshape(-1)]).astype(np.float32)
    assert blob.shape[0] == DEC_FLOATS, blob.shape
    return blob


def pack_decoder_backward(w: Dict[str, np.ndarray]) -> np.ndarray:
    """Transposed decoder layers for the input-gradient chain of `decoder_tile_grad` (mlp.hip.h).  Each W^T is packed like a
    forward layer: rows (MFMA M) = the layer's INPUT features, k = its OUTPUT features in D-fragment order (the masked upstream
    gradient blocks are the B operands)."""
    Ws, bs, Wu, bu = fold_decoder(w)
    parts = [pack_A(np.ascontiguousarray(Ws[3].T), 4, 16, kmap_dfrag),      # (128 in: h2 96 | x0 32) x (128 out)
             pack_A(np.ascontiguousarray(Ws[2].T), 4, 12, kmap_dfrag),      # (128 in) x (96 out)
             pack_A(np.ascontiguousarray(Ws[1].T), 4, 16, kmap_dfrag),      # (128) x (128)
             pack_A(np.ascontiguousarray(Ws[0].T), 1, 16, kmap_dfrag)]      # (32 in: latent 29 | xyz 3) x (128 out)
    blob = np.concatenate([p.reshape(-1) for p in parts]).astype(np.float32)
    assert blob.shape[0] == DECB_FLOATS, blob.shape
    return blob


def pack_decoder_fold(w: Dict[str, np.ndarray]) -> np.ndarray:
    """Latent columns of the two decoder layers that see the input (lin0: x0[:29]; lin3: the skip part, columns 96..124), transposed
    to positions p of the accumulator fragments (p = mb*32 + half*16 + r <-> feature mb*32 + f(r, half)), stored [k][lane][4].
    The latent of a voxel is the same for every sample of that voxel, so  c = bias + W[:, latent] z  is computed once per voxel
    (`decoder_fold_consts`, mlp.hip.h) and used as the accumulator's initial value; only the three coordinate columns are left to the
    MFMAs (2 k-steps per out-block instead of 16)."""
    Ws, bs, Wu, bu = fold_decoder(w)
    out = np.zeros((2, 29, 128), dtype=np.float32)
    for layer, (W, c0) in enumerate(((Ws[0], 0), (Ws[3], 96))):
        for mb in range(4):
            for half in range(2):
                for r in range(16):
                    p = mb * 32 + half * 16 + r
                    f = mb * 32 + _frag_feature(r, half)
                    out[layer, :, p] = W[f, c0:c0 + 29]
    # device order [k][lane][4]: the four positions a lane owns (lin0: lane, lane+64; lin3: lane, lane+64) in one 16-byte load
    dev = np.stack([out[0, :, :64], out[0, :, 64:], out[1, :, :64], out[1, :, 64:]], axis=-1)     # (29, 64, 4)
    blob = np.ascontiguousarray(dev).reshape(-1).astype(np.float32)
    assert blob.shape[0] == DECF_FLOATS
    return blob


# ---- fp32 products on the bf16 matrix pipe ("x6") ----------------------------------------------------------------------------------
# gfx950 runs v_mfma_f32_32x32x16_bf16 at 16x the rate of the f32-input MFMA.  An fp32 value is the exact sum of three bf16 slices
# (8 + 8 + 8 mantissa bits: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid), all differences exact in fp32), every slice
# product is exact in the fp32 accumulator, and the six products whose weight is >= 2^-16 of the leading one
#   hi*hi, hi*mid, mid*hi, hi*lo, mid*mid, lo*hi
# reproduce the fp32 product to ~2^-24 relative — the rounding class of the f32 MFMA itself (measured against float64 on a 128-term
# dot product: 1.05e-6 with the six-slice scheme, 1.42e-6 with v_mfma_f32_32x32x2_f32) — at 6/16 of its matrix-pipe time.

def bf16_rne(x: np.ndarray) -> np.ndarray:
    """float32 -> bf16 bit patterns (uint16), round to nearest even (finite inputs)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def bf16_to_f32(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)


def split_bf16x3(x: np.ndarray):
    x = np.ascontiguousarray(x, dtype=np.float32)
    hi = bf16_rne(x)
    r1 = x - bf16_to_f32(hi)
    mid = bf16_rne(r1)
    r2 = r1 - bf16_to_f32(mid)
    lo = bf16_rne(r2)
    return hi, mid, lo


def pack_A_x6(W: np.ndarray, NMO: int, NKB: int) -> np.ndarray:
    """W (M_out, 32*NKB inputs in D-fragment block order) -> uint16 [kb][s][mo][slice][lane][8]: the A operands of
    v_mfma_f32_32x32x16_bf16 (lane = (row i = lane & 31, half = lane >> 5), element j <-> k = 8*half + j), where k-step s of input
    block kb contracts the features held by accumulator registers 8s..8s+7 of the previous layer's out-block kb."""
    M, K = W.shape
    sl = split_bf16x3(W)
    out = np.zeros((NKB, 2, NMO, 3, 64, 8), dtype=np.uint16)
    for kb in range(NKB):
        for s in range(2):
            for mo in range(NMO):
                for lane in range(64):
                    m = mo * 32 + (lane & 31)
                    if m >= M:
                        continue
                    for j in range(8):
                        k = kb * 32 + _frag_feature(8 * s + j, lane >> 5)
                        if k < K:
                            for q in range(3):
                                out[kb, s, mo, q, lane, j] = sl[q][m, k]
    return out


def pack_decoder_x6(w: Dict[str, np.ndarray]) -> np.ndarray:
    """uint8 blob for the split-bf16 decoder tiles: [aux fp32: A0 coordinate k-group | b0 | b1 | b2 | b3 | sdf head | std head | head
    biases | A3 coordinate k-group] [lin1 slices] [lin2 slices] [lin3 slices (the 96 h2 columns)]."""
    Ws, bs, Wu, bu = fold_decoder(w)

    def kmap_l3(t, half):
        return kmap_dfrag(t, half) if t < 48 else 96 + kmap_natural(t - 48, half)

    a0 = pack_A(Ws[0], 4, 4, kmap_natural)[:, 3]              # (4, 64, 4): features 24..31 of x0, the MFMAs use .z/.w = (28,29),(30,31)
    a3 = pack_A(Ws[3], 4, 16, kmap_l3)[:, 15]
    aux = np.concatenate([a0.reshape(-1), pack_vec(bs[0], 4).reshape(-1), pack_vec(bs[1], 4).reshape(-1), pack_vec(bs[2], 3).reshape(-1),
                          pack_vec(bs[3], 4).reshape(-1), pack_vec(Ws[4][0], 4).reshape(-1), pack_vec(Wu[0], 4).reshape(-1),
                          np.array([bs[4][0], bu[0], 0.0, 0.0], dtype=np.float32), a3.reshape(-1)]).astype(np.float32)
    assert aux.shape[0] == X6_AUX_FLOATS, aux.shape
    l1 = pack_A_x6(Ws[1], 4, 4)
    l2 = pack_A_x6(Ws[2], 3, 4)
    l3 = pack_A_x6(Ws[3][:, :96], 4, 3)
    blob = np.concatenate([aux.view(np.uint8), l1.reshape(-1).view(np.uint8), l2.reshape(-1).view(np.uint8), l3.reshape(-1).view(np.uint8)])
    assert blob.shape[0] == X6_BYTES, blob.shape
    return blob


def pack_encoder_x6(w: Dict[str, np.ndarray]) -> np.ndarray:
    """uint8 blob for `encoder_tile_x6`: [aux fp32: A0 | b0 | b1 | b2 | b3] [lin1 slices] [lin2 / lin3 steps in consumption order:
    L2(0) | L2(1) L3(0) | ... | L2(7) L3(6) | L3(7)]."""
    Ws, bs = fold_encoder(w)
    aux = np.concatenate([pack_A(Ws[0], 1, 1, kmap_natural).reshape(-1), pack_vec(bs[0], 1).reshape(-1), pack_vec(bs[1], 2).reshape(-1),
                          pack_vec(bs[2], 8).reshape(-1), pack_vec(bs[3], 1).reshape(-1)]).astype(np.float32)
    assert aux.shape[0] == E6_AUX_FLOATS, aux.shape
    l1 = pack_A_x6(Ws[1], 2, 1)                                           # (1, 2, 2, 3, 64, 8)
    l2 = pack_A_x6(Ws[2], 8, 2).transpose(2, 0, 1, 3, 4, 5)               # (mb, kb, s, slice, lane, 8)
    l3 = pack_A_x6(Ws[3], 1, 8)[:, :, 0]                                  # (mb, s, slice, lane, 8)
    steps = [l2[0].reshape(-1)]
    for mb in range(8):
        if mb + 1 < 8:
            steps.append(l2[mb + 1].reshape(-1))
        steps.append(l3[mb].reshape(-1))
    blob = np.concatenate([aux.view(np.uint8), np.ascontiguousarray(l1).reshape(-1).view(np.uint8)] +
                          [np.ascontiguousarray(x).view(np.uint8) for x in steps])
    assert blob.shape[0] == E6_BYTES, blob.shape
    return blob


def pack_A_x6_nat(W: np.ndarray, NMO: int) -> np.ndarray:
    """As pack_A_x6 for ONE 32-feature input block that arrives in natural order (B-operand register t of half h holds feature 2t + h:
    the rows of `dif_decode_rows` / the [latent | xyz] input): element j of k-step s <-> feature 2*(8s + j) + half."""
    M, K = W.shape
    assert K == 32
    sl = split_bf16x3(W)
    out = np.zeros((1, 2, NMO, 3, 64, 8), dtype=np.uint16)
    for s in range(2):
        for mo in range(NMO):
            for lane in range(64):
                m = mo * 32 + (lane & 31)
                if m >= M:
                    continue
                for j in range(8):
                    k = 2 * (8 * s + j) + (lane >> 5)
                    for q in range(3):
                        out[0, s, mo, q, lane, j] = sl[q][m, k]
    return out


def pack_decoder_x6u(w: Dict[str, np.ndarray]) -> np.ndarray:
    """The two input blocks the folded tiles do not need: lin0 (all 32 inputs) and the skip block of lin3 (columns 96..127 = x0)."""
    Ws, bs, Wu, bu = fold_decoder(w)
    blob = np.concatenate([pack_A_x6_nat(Ws[0], 4).reshape(-1).view(np.uint8), pack_A_x6_nat(Ws[3][:, 96:128], 4).reshape(-1).view(np.uint8)])
    assert blob.shape[0] == X6U_BYTES, blob.shape
    return blob


def pack_decoder_x6_backward(w: Dict[str, np.ndarray]) -> np.ndarray:
    """bf16 slices of the transposed decoder layers for `decoder_tile_grad_x6` (mlp.hip.h): W3^T | W2^T | W1^T | W0^T, each packed like a
    forward layer — rows (MFMA M) = the layer's INPUT features, k = its OUTPUT features in D-fragment order (the masked upstream
    gradient blocks are the B operands).  Same matrices as pack_decoder_backward."""
    Ws, bs, Wu, bu = fold_decoder(w)
    parts = [pack_A_x6(np.ascontiguousarray(Ws[3].T), 4, 4),      # (128 in: h2 96 | x0 32) x (128 out)
             pack_A_x6(np.ascontiguousarray(Ws[2].T), 4, 3),      # (128 in) x (96 out)
             pack_A_x6(np.ascontiguousarray(Ws[1].T), 4, 4),      # (128) x (128)
             pack_A_x6(np.ascontiguousarray(Ws[0].T), 1, 4)]      # (32 in: latent 29 | xyz 3) x (128 out)
    blob = np.concatenate([p.reshape(-1).view(np.uint8) for p in parts])
    assert blob.shape[0] == X6B_BYTES, blob.shape
    return blob
