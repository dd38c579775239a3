"""The two matrix-pipe variants of the MLP tiles (mlp.hip.h): the default computes every fp32 product as six exact bf16 slice products
on v_mfma_f32_32x32x16_bf16 ("x6"), DIF_DECODER_PIPE=f32 keeps v_mfma_f32_32x32x2_f32.  Both must meet the same parity bars against
the reference goldens / the oracle, and agree with each other to fp32 rounding."""
import numpy as np
import pytest
import torch

from di_fusion_amd import synthetic as syn
from tests.conftest import GOLDEN
from tests import test_gpu_map as TM

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def test_models_really_differ(gpu_model, gpu_model_f32):
    w6, w32 = gpu_model.packed.weights_struct(DEV), gpu_model_f32.packed.weights_struct(DEV)
    assert w6.dec_x6_packed and w6.enc_x6_packed and not w32.dec_x6_packed and not w32.enc_x6_packed


@pytest.mark.parametrize("name", ["seq_small", "seq_c2"])
def test_f32_pipe_sequence_vs_golden_and_oracle(name, gpu_model_f32, oracle_net):
    """tests/test_gpu_map.py runs this on the default pipe; same bars on the f32-input MFMA."""
    TM.test_sequence_vs_golden_and_oracle(name, gpu_model_f32, oracle_net)


def test_encoder_rows_both_pipes_vs_reference_golden(gpu_model, gpu_model_f32):
    g = np.load(GOLDEN / "networks.npz")
    x = torch.from_numpy(g["enc_x"]).to(DEV)
    a = gpu_model.encoder(x).cpu().numpy()
    b = gpu_model_f32.encoder(x).cpu().numpy()
    assert np.abs(a - g["enc_out"]).max() < 2e-5 and np.abs(b - g["enc_out"]).max() < 2e-5
    rel = np.abs(a - b).max() / np.abs(b).max()
    print(f"encoder: x6 vs f32 pipe max diff {np.abs(a - b).max():.2e} (relative to max |y| {rel:.2e})")
    assert rel < 5e-6          # both are fp32-rounding-level perturbations of the same sums (measured 2.9e-6)
    # ragged sizes through the same kernels
    for n in (1, 31, 33, 383):
        xa = x[:n].contiguous()
        assert np.abs(gpu_model.encoder(xa).cpu().numpy() - g["enc_out"][:n]).max() < 2e-5


def test_stream_frames_agree_between_pipes(gpu_model, gpu_model_f32):
    """C2 stream, 4 frames, both pipes: identical integer state, latents / cubes equal to fp32 rounding, same mesh up to the samples
    whose |sdf| sits at the refinement threshold."""
    from di_fusion_amd.system.map import DenseIndexedMap
    scene, cfg = syn.config_c2()
    intr = syn.Intrinsic()
    maps = [DenseIndexedMap(mod, cfg.namespace(), 29, DEV, initial_capacity=16384) for mod in (gpu_model, gpu_model_f32)]
    for f in range(4):
        xyz, nrm = syn.frame_points(scene, f, intr, deg_per_frame=0.5)
        xyz, nrm = xyz.to(DEV), nrm.to(DEV)
        out = []
        for m in maps:
            mask = m.integrate_keyframe(xyz, nrm)
            m.extract_mesh_arrays(4, int(4e6), max_std=0.15, no_cache=False)
            t = m._xbuf[1]
            B = m.last_counters["B"]
            out.append((mask.cpu().numpy(), m.n_occupied, m.latent_vecs[:m.n_occupied].cpu().numpy(), m.voxel_obs_count[:m.n_occupied].cpu().numpy(),
                        t["occ_slot"][:B].cpu().numpy(), t["cube_sdf"][:B].cpu().numpy(), t["cube_std"][:B].cpu().numpy(), dict(m.last_counters)))
        a, b = out
        assert np.array_equal(a[0], b[0]) and a[1] == b[1] and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
        dz = np.abs(a[2] - b[2]).max()
        ds, dd = np.abs(a[5] - b[5]), np.abs(a[6] - b[6])
        # a sample re-decoded on one pipe and interpolated on the other (|sdf| within rounding of 0.05) differs by the interpolation error
        flipped = ds > 1e-4
        print(f"frame {f}: latent maxdiff {dz:.2e}, cube sdf maxdiff {ds[~flipped].max():.2e} std {dd[~flipped].max():.2e}, "
              f"threshold flips {int(flipped.sum())}, VH {a[7]['VH']} / {b[7]['VH']}, T {a[7]['T']} / {b[7]['T']}")
        assert dz < 2e-6
        assert flipped.sum() <= 4 and abs(a[7]["VH"] - b[7]["VH"]) <= 4
        assert ds[~flipped].max() < 3e-5 and dd[~flipped].max() < 3e-5          # fp32 rounding through five layers (either pipe vs the oracle: ~8e-6)
        assert abs(a[7]["T"] - b[7]["T"]) <= 8


def test_repeated_frames_are_bit_identical():
    """tools/determinism_stress.py: frame 0 of the C3 stream 25 times on fresh maps (~3 million MLP tiles in total) — every repeat must
    reproduce the first one bit for bit.  A build with packed-fp32 fold constants failed this within a few repeats."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    import determinism_stress
    bad, counters = determinism_stress.run(25, verbose=True)
    assert counters["B"] > 10000 and bad == 0


def test_decoder_rows_and_point_queries_both_pipes(gpu_model, gpu_model_f32):
    """Explicit (N,32) rows and `get_sdf` values (no gradient) run the unfolded tile: bf16 pipe (`k_decode_x6`) and f32-input MFMA
    (`k_decode`) against the reference golden rows, and against each other on a map's point query."""
    from di_fusion_amd.system.map import DenseIndexedMap
    g = np.load(GOLDEN / "networks.npz")
    x = torch.from_numpy(g["dec_x"]).to(DEV)
    for model in (gpu_model, gpu_model_f32):
        for n in (384, 33, 1):
            sdf, std = model.decoder(x[:n].contiguous())
            assert np.abs(sdf.cpu().numpy() - g["dec_sdf"][:n]).max() < 1e-5 and np.abs(std.cpu().numpy() - g["dec_std"][:n]).max() < 1e-5
    scene, cfg = syn.config_c2()
    xyz, nrm = (t.to(DEV) for t in syn.frame_points(scene, 0, syn.Intrinsic().scaled(0.5), deg_per_frame=0.5))
    out = []
    for model in (gpu_model, gpu_model_f32):
        m = DenseIndexedMap(model, cfg.namespace(), 29, DEV, initial_capacity=16384)
        m.integrate_keyframe(xyz, nrm)
        sdf, std, mask = m.get_sdf(xyz[::7].contiguous())
        out.append((sdf.cpu().numpy(), std.cpu().numpy(), mask.cpu().numpy()))
    assert np.array_equal(out[0][2], out[1][2]) and out[0][2].sum() > 1000
    d = max(np.abs(out[0][0] - out[1][0]).max(), np.abs(out[0][1] - out[1][1]).max())
    print(f"get_sdf values, bf16 pipe vs f32 pipe: max diff {d:.2e} over {int(out[0][2].sum())} points")
    assert d < 2e-5


def test_point_query_gradient_both_pipes(gpu_model, gpu_model_f32):
    """`get_sdf` with `xyz.requires_grad` (the tracker's call): values and the analytic d sdf / d xyz from the bf16-pipe kernel
    (`k_decode_grad_x6`) against the f32-input MFMA kernel (`k_decode<GRAD>`); tests/test_gpu_map.py checks the default pipe against
    float64 autograd and finite differences."""
    from di_fusion_amd.system.map import DenseIndexedMap
    scene, cfg = syn.config_c2()
    xyz, nrm = (t.to(DEV) for t in syn.frame_points(scene, 0, syn.Intrinsic().scaled(0.5), deg_per_frame=0.5))
    out = []
    for model in (gpu_model, gpu_model_f32):
        m = DenseIndexedMap(model, cfg.namespace(), 29, DEV, initial_capacity=16384)
        m.integrate_keyframe(xyz, nrm)
        q = (xyz[::5] + 0.01).contiguous().requires_grad_(True)
        sdf, std, mask = m.get_sdf(q)
        (g,) = torch.autograd.grad((sdf / std.detach()).sum(), q)
        out.append((sdf.detach().cpu().numpy(), std.cpu().numpy(), mask.cpu().numpy(), g.cpu().numpy()))
    assert np.array_equal(out[0][2], out[1][2]) and out[0][2].sum() > 1000
    dv = max(np.abs(out[0][0] - out[1][0]).max(), np.abs(out[0][1] - out[1][1]).max())
    gs = np.abs(out[1][3]).max()
    dg = np.abs(out[0][3] - out[1][3]).max()
    print(f"values max diff {dv:.2e}; gradient max diff {dg:.2e} (max |g| {gs:.2e})")
    assert dv < 2e-5 and dg < 2e-5 * max(1.0, gs)
