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
