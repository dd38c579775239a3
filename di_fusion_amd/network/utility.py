"""Host-side mirror of the reference's `pytorch/network/utility.py` for the fusion path.

Same names and argument meaning: `Networks`, `load_model`, `forward_model`, `get_samples`, `groupby_reduce`.
The decoder / encoder are not `nn.Module`s here: they are handles to the packed weights that the MFMA kernels in
libdifusion.so consume (`csrc/mlp.hip.h`); calling them runs those kernels.  No CPU fallback.
"""
from __future__ import annotations

import argparse
import json
import math
import os
from pathlib import Path
from typing import Dict, Optional

import numpy as np
import torch

from .. import _lib
from . import packing

DEFAULT_WEIGHTS = Path(__file__).resolve().parent / "weights_default.npz"      # ckpt/default of the reference, as arrays


def _x6_default() -> bool:
    """Extract-path decode tiles on the bf16 matrix pipe (mlp.hip.h "x6")?  DIF_DECODER_PIPE=f32 keeps them on the f32-input MFMA."""
    return os.environ.get("DIF_DECODER_PIPE", "bf16x6").lower() != "f32"


class _PackedNet:
    """Packed weights resident on one GPU."""

    def __init__(self, raw: Dict[str, np.ndarray], x6: Optional[bool] = None):
        self.raw = raw
        self.x6 = _x6_default() if x6 is None else bool(x6)
        self._enc_blob = packing.pack_encoder(raw)
        self._dec_blob = packing.pack_decoder(raw)
        self._decb_blob = packing.pack_decoder_backward(raw)
        self._decf_blob = packing.pack_decoder_fold(raw)
        self._x6_blob = packing.pack_decoder_x6(raw) if self.x6 else None
        self._e6_blob = packing.pack_encoder_x6(raw) if self.x6 else None
        self._x6u_blob = packing.pack_decoder_x6u(raw) if self.x6 else None
        self._x6b_blob = packing.pack_decoder_x6_backward(raw) if self.x6 else None
        self._dev = {}

    def weights_struct(self, device: torch.device):
        """(DifWeights, keepalive tensors) for `device`."""
        key = str(device)
        if key not in self._dev:
            enc = torch.from_numpy(self._enc_blob).to(device)
            dec = torch.from_numpy(self._dec_blob).to(device)
            decb = torch.from_numpy(self._decb_blob).to(device)
            decf = torch.from_numpy(self._decf_blob).to(device)
            x6 = torch.from_numpy(self._x6_blob).to(device) if self._x6_blob is not None else None
            e6 = torch.from_numpy(self._e6_blob).to(device) if self._e6_blob is not None else None
            x6u = torch.from_numpy(self._x6u_blob).to(device) if self._x6u_blob is not None else None
            x6b = torch.from_numpy(self._x6b_blob).to(device) if self._x6b_blob is not None else None
            w = _lib.DifWeights(_lib.ptr(enc), enc.numel(), _lib.ptr(dec), dec.numel(), _lib.ptr(decb), decb.numel(), _lib.ptr(decf), decf.numel(),
                                _lib.ptr(x6) if x6 is not None else None, x6.numel() if x6 is not None else 0,
                                _lib.ptr(e6) if e6 is not None else None, e6.numel() if e6 is not None else 0,
                                _lib.ptr(x6u) if x6u is not None else None, x6u.numel() if x6u is not None else 0,
                                _lib.ptr(x6b) if x6b is not None else None, x6b.numel() if x6b is not None else 0)
            self._dev[key] = (w, enc, dec, decb, decf, x6, e6, x6u, x6b)
        return self._dev[key][0]


class Decoder:
    """Callable like the reference's `di_decoder.Model` in eval mode: (N,32) -> (sdf (N,1), std (N,1))
    (reference `network/di_decoder.py:55-86`)."""

    def __init__(self, packed: _PackedNet):
        self.packed = packed

    def eval(self):
        return self

    def __call__(self, x: torch.Tensor):
        _lib.require_cuda(x)
        if x.dim() != 2 or x.size(1) != _lib.LATENT_DIM + 3 or x.dtype != torch.float32:
            raise RuntimeError("decoder input must be (N, 32) float32")
        n = x.size(0)
        sdf = torch.empty((n, 1), dtype=torch.float32, device=x.device)
        std = torch.empty((n, 1), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            w = self.packed.weights_struct(x.device)
            _lib.check(_lib.load().dif_decode_rows(w, _lib.ptr(x), n, _lib.ptr(sdf), _lib.ptr(std), _lib.stream_ptr()),
                       "dif_decode_rows")
        return sdf, std


class Encoder:
    """Callable like `di_encoder.Model(mode='cnp')`: (N,6) -> (N,29) (reference `network/di_encoder.py:26-30`)."""

    def __init__(self, packed: _PackedNet):
        self.packed = packed

    def eval(self):
        return self

    def __call__(self, x: torch.Tensor):
        _lib.require_cuda(x)
        if x.dim() != 2 or x.size(1) != 6 or x.dtype != torch.float32:
            raise RuntimeError("encoder input must be (N, 6) float32")
        n = x.size(0)
        out = torch.empty((n, _lib.LATENT_DIM), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            w = self.packed.weights_struct(x.device)
            _lib.check(_lib.load().dif_encode_rows(w, _lib.ptr(x), n, _lib.ptr(out), _lib.stream_ptr()), "dif_encode_rows")
        return out


class Networks:
    """reference `network/utility.py:10-19`."""

    def __init__(self):
        self.decoder: Optional[Decoder] = None
        self.encoder: Optional[Encoder] = None
        self.packed: Optional[_PackedNet] = None

    def eval(self):
        return self


def networks_from_arrays(raw: Dict[str, np.ndarray], x6: Optional[bool] = None) -> Networks:
    """x6: run the MLP tiles on the bf16 matrix pipe (None: the DIF_DECODER_PIPE default, see _x6_default)."""
    packed = _PackedNet({k: np.asarray(v) for k, v in raw.items()}, x6)
    m = Networks()
    m.packed = packed
    m.decoder = Decoder(packed)
    m.encoder = Encoder(packed)
    return m


def random_weights(seed: int = 0) -> Dict[str, np.ndarray]:
    """Random-init weights of the shipped architecture (hyper.json:34-63) for runs without a checkpoint."""
    g = np.random.default_rng(seed)
    raw = {}
    dims = [(128, 32), (128, 128), (96, 128), (128, 128), (1, 128)]
    for i, (o, k) in enumerate(dims):
        raw[f"decoder.lin{i}.weight_v"] = (g.standard_normal((o, k)) / math.sqrt(k)).astype(np.float32)
        raw[f"decoder.lin{i}.weight_g"] = np.ones((o, 1), dtype=np.float32)
        raw[f"decoder.lin{i}.bias"] = (g.standard_normal(o) * 0.01).astype(np.float32)
    raw["decoder.uncertainty_layer.weight"] = (g.standard_normal((1, 128)) / math.sqrt(128)).astype(np.float32)
    raw["decoder.uncertainty_layer.bias"] = np.zeros(1, dtype=np.float32)
    feats = [6, 32, 64, 256]
    for i in range(3):
        raw[f"encoder.mlp.layer{i}.conv.weight"] = (g.standard_normal((feats[i + 1], feats[i], 1)) * math.sqrt(2.0 / feats[i])).astype(np.float32)
        raw[f"encoder.mlp.layer{i}.normlayer.bn.weight"] = np.ones(feats[i + 1], dtype=np.float32)
        raw[f"encoder.mlp.layer{i}.normlayer.bn.bias"] = np.zeros(feats[i + 1], dtype=np.float32)
        raw[f"encoder.mlp.layer{i}.normlayer.bn.running_mean"] = np.zeros(feats[i + 1], dtype=np.float32)
        raw[f"encoder.mlp.layer{i}.normlayer.bn.running_var"] = np.ones(feats[i + 1], dtype=np.float32)
    raw["encoder.mlp.layer3.conv.weight"] = (g.standard_normal((29, 256, 1)) * math.sqrt(1.0 / 256)).astype(np.float32)
    raw["encoder.mlp.layer3.conv.bias"] = np.zeros(29, dtype=np.float32)
    return raw


def load_weights_npz(path=DEFAULT_WEIGHTS) -> Dict[str, np.ndarray]:
    return {k: v for k, v in np.load(path).items()}


def load_model(training_hyper_path: str, use_epoch: int = -1):
    """reference `network/utility.py:22-58`: hyper.json next to `model_<epoch>.pth.tar` / `encoder_<epoch>.pth.tar`
    -> (Networks, args).  Only the shipped architecture (di_decoder + di_encoder, hyper.json:34-63) is supported."""
    training_hyper_path = Path(training_hyper_path)
    if training_hyper_path.suffix != ".json":
        raise NotImplementedError("only trained checkpoints (hyper.json) are supported on the fusion path")
    args = argparse.Namespace(**json.loads(training_hyper_path.read_text()))
    exp_dir = training_hyper_path.parent
    model_paths = {int(str(t).split("model_")[-1].split(".pth")[0]): t for t in exp_dir.glob("model_*.pth.tar")}
    if use_epoch == -1 and model_paths:
        use_epoch = max(model_paths)
    assert use_epoch in model_paths.keys(), f"{use_epoch} not found in {sorted(list(model_paths.keys()))}"
    args.checkpoint = model_paths[use_epoch]
    if args.network_name != "di_decoder" or args.encoder_name != "di_encoder" or args.code_length != _lib.LATENT_DIM:
        raise NotImplementedError("libdifusion is specialised to the shipped di_decoder/di_encoder topology")
    raw = {}
    for k, v in torch.load(args.checkpoint, map_location="cpu")["model_state"].items():
        raw["decoder." + k] = v.detach().cpu().numpy()
    for k, v in torch.load(exp_dir / f"encoder_{use_epoch}.pth.tar", map_location="cpu")["model_state"].items():
        raw["encoder." + k] = v.detach().cpu().numpy()
    # A weight-normed Linear pickled with its hook-computed `weight` still attached (what the reference's
    # `fix_weight_norm_pickle`, utility.py:211-220, strips before pickling) carries a stale copy next to weight_g / weight_v:
    # g and v are the parameters, the copy is dropped.  (`packing.fold_decoder` folds g * v / |v|; a checkpoint trained without
    # weight_norm has only `weight` and is used as is.)
    for k in [k for k in raw if k.endswith(".weight") and k[:-len("weight")] + "weight_v" in raw]:
        del raw[k]
    return networks_from_arrays(raw), args


def forward_model(model, network_input: torch.Tensor = None, latent_input: torch.Tensor = None,
                  xyz_input: torch.Tensor = None, loss_func=None, max_sample: int = 2 ** 32,
                  no_detach: bool = False, verbose: bool = False):
    """reference `network/utility.py:61-126`, inference part (loss_func / autograd are the optimiser path, out of scope)."""
    if loss_func is not None:
        raise NotImplementedError("forward_model(loss_func=...) belongs to the latent-optimisation path (SURVEY.md 8f-4)")
    if latent_input is not None and xyz_input is not None:
        assert network_input is None
        network_input = torch.cat((latent_input, xyz_input), dim=1)
    assert network_input.ndimension() == 2
    n_chunks = max(1, math.ceil(network_input.size(0) / max_sample))
    assert not no_detach or n_chunks == 1
    outs = [model(c.contiguous()) for c in torch.chunk(network_input, n_chunks)]
    return [torch.cat([o[i] for o in outs], dim=0) for i in range(2)]


def get_samples(r: int, device: torch.device, a: float = 0.0, b: float = None):
    """The sample lattice of reference `network/utility.py:129-149`: (r^3, 3) float32, x slowest, coordinate fl(fl(i) * vsize) + a per
    axis (the kernels generate the same values in registers, `csrc/kernels_extract.hip.h:Lattice`)."""
    r = int(r)
    if b is None:
        b = 1. - 1. / r
    axis = torch.arange(r, device=device, dtype=torch.float32) * ((b - a) / (r - 1)) + a
    return torch.stack(torch.meshgrid(axis, axis, axis, indexing="ij"), dim=-1).reshape(-1, 3)


def groupby_reduce(sample_indexer: torch.Tensor, sample_values: torch.Tensor, op: str = "max"):
    """reference `network/utility.py:186-208` on top of `dif_groupby_sum`: per-group 'sum' or 'mean' of the rows of `sample_values`
    (groups 0 .. max index); any other op raises, as there."""
    from ..system.ext import groupby_sum
    if op not in ("sum", "mean"):
        raise NotImplementedError
    assert sample_indexer.size(0) == sample_values.size(0), "Indexer and Values must agree on sample count!"
    total, count = groupby_sum(sample_values, sample_indexer, int(sample_indexer.max().item()) + 1)
    return total if op == "sum" else total / count.unsqueeze(-1)
