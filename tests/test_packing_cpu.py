"""CPU: the packed-weight layout + MFMA chain convention reproduce the oracle networks (emulated lane by lane)."""
import numpy as np

from di_fusion_amd.network import packing as P
from tests import mfma_emulator as E
from oracle import difusion_oracle as O
from tests.conftest import GOLDEN


def test_blob_sizes(raw_weights):
    assert P.pack_encoder(raw_weights).shape == (P.ENC_FLOATS,)
    assert P.pack_decoder(raw_weights).shape == (P.DEC_FLOATS,)


def test_encoder_chain_matches_oracle(raw_weights, oracle_net):
    blob = P.pack_encoder(raw_weights).astype(np.float64)
    x = np.load(GOLDEN / "networks.npz")["enc_x"][:32]
    got = E.encoder_tile(blob, x.astype(np.float64))
    want = oracle_net.encoder(x)
    assert np.abs(got - want).max() < 2e-5


def test_decoder_chain_matches_oracle(raw_weights, oracle_net):
    blob = P.pack_decoder(raw_weights).astype(np.float64)
    g = np.load(GOLDEN / "networks.npz")
    x = g["dec_x"][:32]
    ps, pu = E.decoder_tile(blob, x.astype(np.float64))
    sdf = np.tanh(ps)
    std = 0.05 + 0.5 * np.log1p(np.exp(pu))
    assert np.abs(sdf - g["dec_sdf"][:32, 0]).max() < 2e-5
    assert np.abs(std - g["dec_std"][:32, 0]).max() < 2e-5


def test_folded_decoder_chain_matches_oracle(raw_weights, oracle_net):
    """The per-voxel constant folding (packing.pack_decoder_fold + the 2-k-step coordinate MFMAs) computes the same decoder."""
    blob = P.pack_decoder(raw_weights).astype(np.float64)
    fold = P.pack_decoder_fold(raw_weights).astype(np.float64)
    assert fold.shape[0] == P.DECF_FLOATS
    g = np.random.default_rng(5)
    latent = (g.normal(size=29) * 0.5).astype(np.float32)
    pts = (g.random((32, 3)) - 0.5).astype(np.float32)
    ps, pu = E.decoder_tile_folded(blob, fold, latent.astype(np.float64), pts.astype(np.float64))
    rows = np.concatenate([np.repeat(latent[None], 32, 0), pts], 1).astype(np.float32)
    # against the unfolded emulated chain (same packing conventions) and against the oracle's forward
    ps0, pu0 = E.decoder_tile(blob, rows.astype(np.float64))
    assert np.abs(ps - ps0).max() < 1e-9 and np.abs(pu - pu0).max() < 1e-9
    sdf, std = O.forward_model(oracle_net, rows[:, :29], rows[:, 29:])
    assert np.abs(np.tanh(ps) - sdf.reshape(-1)).max() < 2e-5


def test_bf16_slices_are_exact_and_rne():
    """An fp32 value is the exact sum of its three bf16 slices, and the first slice is torch's round-to-nearest-even bf16."""
    import torch
    g = np.random.default_rng(11)
    x = np.concatenate([g.normal(size=4000), g.normal(size=2000) * 1e-6, g.normal(size=2000) * 1e6, [0.0, -0.0, 1.0, -1.0, 3.0e38]]).astype(np.float32)
    hi, mid, lo = P.split_bf16x3(x)
    assert np.array_equal(hi, torch.from_numpy(x).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16))
    total = P.bf16_to_f32(hi).astype(np.float64) + P.bf16_to_f32(mid).astype(np.float64) + P.bf16_to_f32(lo).astype(np.float64)
    assert np.array_equal(total.astype(np.float32), x) and np.abs(total - x.astype(np.float64)).max() == 0.0


def test_x6_blob_sizes(raw_weights):
    assert P.pack_decoder_x6(raw_weights).shape == (P.X6_BYTES,) and P.X6_BYTES == 256912
    assert P.pack_encoder_x6(raw_weights).shape == (P.E6_BYTES,) and P.E6_BYTES == 162304 <= 160 * 1024


def test_x6_decoder_chain_matches_f32_chain_and_oracle(raw_weights, oracle_net):
    """The bf16-sliced folded decoder (packing.pack_decoder_x6 + six slice products per k-step) against the f32-MFMA chain
    emulation and the oracle: dropping the three smallest slice products leaves ~2^-24 relative per product."""
    blob = P.pack_decoder(raw_weights).astype(np.float64)
    fold = P.pack_decoder_fold(raw_weights).astype(np.float64)
    x6 = P.pack_decoder_x6(raw_weights)
    g = np.random.default_rng(6)
    for _ in range(2):
        latent = (g.normal(size=29) * 0.5).astype(np.float32)
        pts = (g.random((32, 3)) - 0.5).astype(np.float32)
        ps, pu = E.decoder_tile_folded_x6(x6, fold, latent.astype(np.float64), pts.astype(np.float64))
        ps0, pu0 = E.decoder_tile_folded(blob, fold, latent.astype(np.float64), pts.astype(np.float64))
        # (the x6 emulation also rounds the activations to fp32 between layers, as the kernel does; the f32 chain emulation runs in float64)
        for a, b in ((ps, ps0), (pu, pu0)):
            assert (np.abs(a - b) / np.maximum(1.0, np.abs(b))).max() < 5e-7, (np.abs(a - b) / np.maximum(1.0, np.abs(b))).max()
        rows = np.concatenate([np.repeat(latent[None], 32, 0), pts], 1).astype(np.float32)
        sdf, std = O.forward_model(oracle_net, rows[:, :29], rows[:, 29:])
        assert np.abs(np.tanh(ps) - sdf.reshape(-1)).max() < 2e-5
        assert np.abs(0.05 + 0.5 * np.log1p(np.exp(pu)) - std.reshape(-1)).max() < 2e-5


def test_x6_encoder_chain_matches_f32_chain_and_oracle(raw_weights, oracle_net):
    blob = P.pack_encoder(raw_weights).astype(np.float64)
    e6 = P.pack_encoder_x6(raw_weights)
    x = np.load(GOLDEN / "networks.npz")["enc_x"][:32]
    got = E.encoder_tile_x6(e6, x.astype(np.float64))
    ref = E.encoder_tile(blob, x.astype(np.float64))
    assert np.abs(got - ref).max() < 2e-6 * max(1.0, np.abs(ref).max()), np.abs(got - ref).max()
    assert np.abs(got - oracle_net.encoder(x)).max() < 2e-5
