"""numpy emulation of the v_mfma_f32_32x32x2_f32 chain in `di_fusion_amd/csrc/mlp.hip.h`, lane for lane.

Used on CPU (no GPU needed) to check that the weight packing of `network/packing.py` and the k-order convention of
"transposed chaining" reproduce the encoder / decoder of the oracle.  Fragment layouts are the ones documented in
/opt/skills/guides/cdna_hip_programming.md section 3: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31],
D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5).
"""
import numpy as np

from di_fusion_amd.network import packing as P

LANES = np.arange(64)
ROW = np.array([[(r & 3) + 8 * (r >> 2) + 4 * h for r in range(16)] for h in range(2)])   # [half][reg]


def mfma(a, b, acc):
    """a, b: (64,) per-lane operands; acc: (64,16) per-lane accumulators. float64 accumulate is fine for a layout check."""
    A = np.zeros((32, 2)); B = np.zeros((2, 32))
    A[LANES & 31, LANES >> 5] = a
    B[LANES >> 5, LANES & 31] = b
    D = A @ B
    out = acc.copy()
    for l in range(64):
        out[l] += D[ROW[l >> 5], l & 31]
    return out


def bias16(blob, off):
    out = np.zeros((64, 16))
    for l in range(64):
        out[l] = blob[off + (l >> 5) * 16: off + (l >> 5) * 16 + 16]
    return out


def block_mm(blob, a_off, hin, acc):
    """hin: list of (64,16) blocks; a_off: float offset of k-group 0 of this out-block."""
    for kb, h in enumerate(hin):
        for g in range(4):
            t = kb * 4 + g
            a4 = blob[a_off + t * 256: a_off + (t + 1) * 256].reshape(64, 4)
            for j in range(4):
                acc = mfma(a4[:, j], h[:, 4 * g + j], acc)
    return acc


ENC = dict(A0=0, B0=256, A1=288, B1=2336, A2=2400, B2=18784, A3=19040, B3=27232)
DEC = dict(A0=0, B0=4096, A1=4224, B1=20608, A2=20736, B2=33024, B3=33120, HW=33248, HU=33376, HB=33504, A3=33508)


def encoder_tile(blob, pts):
    """pts (32,6) -> (32,29)"""
    half = LANES >> 5
    col = LANES & 31
    x0 = np.where(half == 1, pts[col, 1], pts[col, 0])
    x1 = np.where(half == 1, pts[col, 3], pts[col, 2])
    x2 = np.where(half == 1, pts[col, 5], pts[col, 4])
    acc = bias16(blob, ENC["B0"])
    a = blob[ENC["A0"]:ENC["A0"] + 256].reshape(64, 4)
    for j, x in enumerate((x0, x1, x2)):
        acc = mfma(a[:, j], x, acc)
    h0 = np.maximum(acc, 0)
    h1 = []
    for mb in range(2):
        acc = block_mm(blob, ENC["A1"] + mb * 4 * 256, [h0], bias16(blob, ENC["B1"] + mb * 32))
        h1.append(np.maximum(acc, 0))
    out = bias16(blob, ENC["B3"])
    for mb in range(8):
        acc = block_mm(blob, ENC["A2"] + mb * 8 * 256, h1, bias16(blob, ENC["B2"] + mb * 32))
        out = block_mm(blob, ENC["A3"] + mb * 4 * 256, [np.maximum(acc, 0)], out)
    res = np.zeros((32, 32))
    for l in range(64):
        res[l & 31, ROW[l >> 5]] = out[l]
    return res[:, :29]


def decoder_tile(blob, rows):
    """rows (32,32) -> pre-activation (sdf_lin (32,), std_lin (32,))"""
    half = LANES >> 5
    col = LANES & 31
    xin = np.zeros((64, 16))
    for t in range(16):
        xin[:, t] = rows[col, 2 * t + half]
    h0 = [np.maximum(block_mm(blob, DEC["A0"] + mb * 4 * 256, [xin], bias16(blob, DEC["B0"] + mb * 32)), 0) for mb in range(4)]
    h1 = [np.maximum(block_mm(blob, DEC["A1"] + mb * 16 * 256, h0, bias16(blob, DEC["B1"] + mb * 32)), 0) for mb in range(4)]
    h2 = [np.maximum(block_mm(blob, DEC["A2"] + mb * 16 * 256, h1, bias16(blob, DEC["B2"] + mb * 32)), 0) for mb in range(3)]
    h2x = h2 + [xin]
    ps = np.zeros(64); pu = np.zeros(64)
    for mb in range(4):
        acc = np.maximum(block_mm(blob, DEC["A3"] + mb * 16 * 256, h2x, bias16(blob, DEC["B3"] + mb * 32)), 0)
        ps += (acc * bias16(blob, DEC["HW"] + mb * 32)).sum(1)
        pu += (acc * bias16(blob, DEC["HU"] + mb * 32)).sum(1)
    ps = ps + ps[LANES ^ 32] + blob[DEC["HB"]]
    pu = pu + pu[LANES ^ 32] + blob[DEC["HB"] + 1]
    return ps[:32], pu[:32]


def decoder_tile_folded(blob, fold_blob, latent, pts):
    """The folded chain of mlp.hip.h (`decoder_fold_consts` + `decoder_tile_folded`): one voxel latent (29,), 32 sample coordinates
    pts (32,3) -> pre-activation (sdf_lin (32,), std_lin (32,)).  fold_blob is packing.pack_decoder_fold's [k][lane][4] array."""
    half = LANES >> 5
    col = LANES & 31
    wk = fold_blob.reshape(29, 64, 4)
    # c[p], p = accumulator-fragment position: lane owns p = lane, lane + 64 of each layer
    c = np.zeros((2, 128))
    c[0, :64] = blob[DEC["B0"]:DEC["B0"] + 64]; c[0, 64:] = blob[DEC["B0"] + 64:DEC["B0"] + 128]
    c[1, :64] = blob[DEC["B3"]:DEC["B3"] + 64]; c[1, 64:] = blob[DEC["B3"] + 64:DEC["B3"] + 128]
    for k in range(29):
        c[0, :64] += wk[k, :, 0] * latent[k]; c[0, 64:] += wk[k, :, 1] * latent[k]
        c[1, :64] += wk[k, :, 2] * latent[k]; c[1, 64:] += wk[k, :, 3] * latent[k]
    b14 = np.where(half == 1, pts[col, 0], 0.0)
    b15 = np.where(half == 1, pts[col, 2], pts[col, 1])
    h0 = []
    for mb in range(4):
        acc = bias16(c[0], mb * 32)
        a4 = blob[DEC["A0"] + (mb * 4 + 3) * 256: DEC["A0"] + (mb * 4 + 4) * 256].reshape(64, 4)
        acc = mfma(a4[:, 2], b14, acc)
        acc = mfma(a4[:, 3], b15, acc)
        h0.append(np.maximum(acc, 0))
    h1 = [np.maximum(block_mm(blob, DEC["A1"] + mb * 16 * 256, h0, bias16(blob, DEC["B1"] + mb * 32)), 0) for mb in range(4)]
    h2 = [np.maximum(block_mm(blob, DEC["A2"] + mb * 16 * 256, h1, bias16(blob, DEC["B2"] + mb * 32)), 0) for mb in range(3)]
    ps = np.zeros(64); pu = np.zeros(64)
    for mb in range(4):
        acc = block_mm(blob, DEC["A3"] + mb * 16 * 256, h2, bias16(c[1], mb * 32))
        a4 = blob[DEC["A3"] + (mb * 16 + 15) * 256: DEC["A3"] + (mb * 16 + 16) * 256].reshape(64, 4)
        acc = mfma(a4[:, 2], b14, acc)
        acc = mfma(a4[:, 3], b15, acc)
        acc = np.maximum(acc, 0)
        ps += (acc * bias16(blob, DEC["HW"] + mb * 32)).sum(1)
        pu += (acc * bias16(blob, DEC["HU"] + mb * 32)).sum(1)
    ps = ps + ps[LANES ^ 32] + blob[DEC["HB"]]
    pu = pu + pu[LANES ^ 32] + blob[DEC["HB"] + 1]
    return ps[:32], pu[:32]


# ---- the bf16-sliced chain ("x6", mlp.hip.h) ------------------------------------------------------------------------------------
# v_mfma_f32_32x32x16_bf16: A[i = l & 31][k = 8*(l >> 5) + j], B[k = 8*(l >> 5) + j][n = l & 31], j = 0..7; D as above.

def mfma_bf16(a8, b8, acc):
    """a8, b8: (64, 8) per-lane operand values (already bf16-representable); acc (64, 16)."""
    A = np.zeros((32, 16)); B = np.zeros((16, 32))
    for j in range(8):
        A[LANES & 31, 8 * (LANES >> 5) + j] = a8[:, j]
        B[8 * (LANES >> 5) + j, LANES & 31] = b8[:, j]
    D = A @ B
    out = acc.copy()
    for l in range(64):
        out[l] += D[ROW[l >> 5], l & 31]
    return out


def slices_of(x):
    """(…) float -> three arrays of the bf16 slice VALUES (hi, mid, lo), as the kernel's split_pair computes them."""
    return [P.bf16_to_f32(s).astype(np.float64) for s in P.split_bf16x3(np.asarray(x, dtype=np.float32))]


def frag_x6(blob_u8, byte_off):
    """The three weight fragments of one step: (3, 64, 8) slice values."""
    raw = blob_u8[byte_off: byte_off + 3072].view(np.uint16).reshape(3, 64, 8)
    return P.bf16_to_f32(raw).astype(np.float64)


def step_x6(a, x, acc):
    """The six slice products the kernel issues (mlp.hip.h:step_x6), smallest first."""
    for qa, qx in ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)):
        acc = mfma_bf16(a[qa], x[qx], acc)
    return acc


def layer_x6(blob_u8, byte_off, hin, kb0, kb1, accs):
    """steps in memory order [kb][s][mo]; hin: list of (64,16) fp32-valued blocks; accs: list of NMO accumulators (updated copies returned)."""
    accs = [a.copy() for a in accs]
    nmo = len(accs)
    t = 0
    for kb in range(kb0, kb1):
        h = hin[kb].astype(np.float32)
        for s in range(2):
            x = [sl[:, 8 * s: 8 * s + 8] for sl in slices_of(h)]
            for mo in range(nmo):
                accs[mo] = step_x6(frag_x6(blob_u8, byte_off + t * 3072), x, accs[mo])
                t += 1
    return accs


def decoder_tile_folded_x6(x6_blob, fold_blob, latent, pts):
    """`decoder_fold_consts_x6` + `decoder_tile_folded_x6`: one voxel latent (29,), pts (32,3) -> pre-activation (sdf_lin, std_lin)."""
    aux = x6_blob[:P.X6_AUX_FLOATS * 4].view(np.float32).astype(np.float64)
    A0C, B0, B1, B2, B3, HW, HU, HB, A3C = 0, 1024, 1152, 1280, 1376, 1504, 1632, 1760, 1764
    L1 = P.X6_AUX_FLOATS * 4
    L2 = L1 + P.X6_L1_BYTES
    L3 = L2 + P.X6_L2_BYTES
    half = LANES >> 5
    col = LANES & 31
    wk = fold_blob.reshape(29, 64, 4)
    c = np.zeros((2, 128))
    c[0] = aux[B0:B0 + 128]; c[1] = aux[B3:B3 + 128]
    for k in range(29):
        c[0, :64] += wk[k, :, 0] * latent[k]; c[0, 64:] += wk[k, :, 1] * latent[k]
        c[1, :64] += wk[k, :, 2] * latent[k]; c[1, 64:] += wk[k, :, 3] * latent[k]
    b14 = np.where(half == 1, pts[col, 0], 0.0)
    b15 = np.where(half == 1, pts[col, 2], pts[col, 1])
    h0 = []
    for mb in range(4):
        a4 = aux[A0C + mb * 256: A0C + (mb + 1) * 256].reshape(64, 4)
        acc = mfma(a4[:, 2], b14, bias16(c[0], mb * 32))
        acc = mfma(a4[:, 3], b15, acc)
        h0.append(np.maximum(acc, 0))
    h1 = [np.maximum(a, 0) for a in layer_x6(x6_blob, L1, h0, 0, 4, [bias16(aux, B1 + mb * 32) for mb in range(4)])]
    h2 = layer_x6(x6_blob, L2, h1, 0, 4, [bias16(aux, B2 + mb * 32) for mb in range(3)])
    h2 = [np.maximum(a, 0) for a in h2]
    h3 = layer_x6(x6_blob, L3, h2, 0, 3, [bias16(c[1], mb * 32) for mb in range(4)])
    ps = np.zeros(64); pu = np.zeros(64)
    for mb in range(4):
        a4 = aux[A3C + mb * 256: A3C + (mb + 1) * 256].reshape(64, 4)
        acc = mfma(a4[:, 2], b14, h3[mb])
        acc = np.maximum(mfma(a4[:, 3], b15, acc), 0)
        ps += (acc * bias16(aux, HW + mb * 32)).sum(1)
        pu += (acc * bias16(aux, HU + mb * 32)).sum(1)
    ps = ps + ps[LANES ^ 32] + aux[HB]
    pu = pu + pu[LANES ^ 32] + aux[HB + 1]
    return ps[:32], pu[:32]


def encoder_tile_x6(e6_blob, pts):
    """`encoder_tile_x6`: pts (32,6) -> (32,29); weight steps consumed in the blob's own order."""
    aux = e6_blob[:P.E6_AUX_FLOATS * 4].view(np.float32).astype(np.float64)
    A0, B0, B1, B2, B3 = 0, 256, 288, 352, 608
    L1 = P.E6_AUX_FLOATS * 4
    L23 = L1 + 12288
    half = LANES >> 5
    col = LANES & 31
    xs = [np.where(half == 1, pts[col, 2 * j + 1], pts[col, 2 * j]) for j in range(3)]
    acc = bias16(aux, B0)
    a = aux[A0:A0 + 256].reshape(64, 4)
    for j in range(3):
        acc = mfma(a[:, j], xs[j], acc)
    h0 = np.maximum(acc, 0)
    h1 = [np.maximum(a_, 0) for a_ in layer_x6(e6_blob, L1, [h0], 0, 1, [bias16(aux, B1), bias16(aux, B1 + 32)])]
    t = [0]

    def l2_block(mb):
        acc = layer_x6(e6_blob, L23 + t[0] * 3072, h1, 0, 2, [bias16(aux, B2 + mb * 32)])[0]
        t[0] += 4
        return acc

    out = bias16(aux, B3)
    cur = l2_block(0)
    for mb in range(8):
        h = np.maximum(cur, 0)
        if mb + 1 < 8:
            cur = l2_block(mb + 1)
        out = layer_x6(e6_blob, L23 + t[0] * 3072, [h], 0, 1, [out])[0]
        t[0] += 2
    assert t[0] == 48
    res = np.zeros((32, 32))
    for l in range(64):
        res[l & 31, ROW[l >> 5]] = out[l]
    return res[:, :29]
