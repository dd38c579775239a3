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
