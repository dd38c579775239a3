"""CPU: the packed-weight layout + MFMA chain convention reproduce the oracle networks (emulated lane by lane)."""
import numpy as np

from di_fusion_amd.network import packing as P
from tests import mfma_emulator as E
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
