#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by IMPORTING the reference on CPU.

Runs only in the build container (needs /root/reference); nothing here travels to the GPU box except
the .npz files it writes.  The reference's own tests pin nothing, so these vectors —
outputs of the reference's Python (`pytorch/system/map.py`, `pytorch/network/*`) on seeded synthetic
inputs — are what pins the oracle under /oracle.

Stubs injected before import:
  * `open3d`  — empty module (mesh container only, `map.py:6,521-543`)
  * `numba`   — identity `jit` (`map.py:20`)
  * `system.ext` — `groupby_sum` = index_add_ (stands in for `ext/indexing/indexing.cu:59-109`) and a
    RECORDING `marching_cubes_interp` (the CUDA kernel cannot run here; the tuple it receives is the golden
    vector for everything up to marching cubes)
  * `torch.cuda.Stream/stream/synchronize` — no-ops (`map.py:232,625-626`)
  * `np.product` — removed in NumPy 2 (`map.py:201,407`)
  * (mesh-cache fixture only, `--only mesh_cache`) `marching_cubes_interp` = the oracle's C restatement, `_get_valid_idx` handed a sentinel:
    see mesh_cache_fixture
  * (tracker fixtures only, `--only track`) `pyquaternion` — the image has none: a `Quaternion` that keeps the rotation as a 3x3 matrix in
    float64 (`rotation_matrix`, `inverse`, `rotate`, `*`), enough for `utils/motion_util.Isometry` to run as written; the rest of
    `system.ext` the tracker imports (CUDA image / point-cloud kernels) as names that are never called; `Tensor.cuda()` = identity
    (`tracker.py:196`)

Usage:  python tests/golden/make_golden.py            (writes *.npz next to this file)
"""
import contextlib
import hashlib
import json
import sys
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
REF = Path("/root/reference/pytorch")
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REF))

torch.manual_seed(0)
np.product = np.prod  # noqa

# ---- stubs -------------------------------------------------------------------------------------
sys.modules["open3d"] = types.ModuleType("open3d")
_numba = types.ModuleType("numba")
_numba.jit = lambda f=None, **kw: (f if f is not None else (lambda g: g))
sys.modules["numba"] = _numba

RECORDED = {}


def _groupby_sum(values, indices, C):
    C = int(C)
    s = torch.zeros((C, values.size(1)), dtype=torch.float32)
    s.index_add_(0, indices, values)
    c = torch.zeros((C,), dtype=torch.int32)
    c.index_add_(0, indices, torch.ones_like(indices, dtype=torch.int32))
    return [s, c]


def _mc_record(indexer, valid_blocks, vec_batch_mapping, cube_sdf, cube_std, max_n, n_xyz, max_std):
    RECORDED["mc_args"] = dict(indexer=indexer.clone(), valid_blocks=valid_blocks.clone(),
                               vec_batch_mapping=vec_batch_mapping.clone(), cube_sdf=cube_sdf.clone(),
                               cube_std=cube_std.clone(), max_n_triangles=int(max_n), n_xyz=list(n_xyz),
                               max_std=float(max_std))
    return (torch.zeros((0, 3, 3)), torch.zeros((0,), dtype=torch.long), torch.zeros((0, 3)))


_ext = types.ModuleType("system.ext")
_ext.groupby_sum = _groupby_sum
_ext.marching_cubes_interp = _mc_record
_system = types.ModuleType("system")
_system.__path__ = [str(REF / "system")]
_system.ext = _ext
sys.modules["system"] = _system
sys.modules["system.ext"] = _ext


class _FakeStream:
    def __init__(self, *a, **k):
        pass

    def synchronize(self):
        pass


torch.cuda.Stream = _FakeStream
torch.cuda.stream = lambda s: contextlib.nullcontext()
torch.cuda.synchronize = lambda *a, **k: None

from system import map as ref_map                     # noqa: E402  (the reference)
from network import di_decoder, di_encoder, utility as ref_util   # noqa: E402
from di_fusion_amd import synthetic as syn            # noqa: E402


def load_reference_model():
    hyper = json.loads((REF / "ckpt/default/hyper.json").read_text())
    model = ref_util.Networks()
    model.decoder = di_decoder.Model(hyper["code_length"], **hyper["network_specs"])
    model.encoder = di_encoder.Model(**hyper["encoder_specs"])
    model.decoder.load_state_dict(torch.load(REF / "ckpt/default/model_300.pth.tar", map_location="cpu")["model_state"])
    model.encoder.load_state_dict(torch.load(REF / "ckpt/default/encoder_300.pth.tar", map_location="cpu")["model_state"])
    model.eval()
    return model, hyper


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def export_weights(model):
    """Raw checkpoint tensors -> di_fusion_amd/network/weights_default.npz (data, not code): the
    package's default weights, which the tests also use."""
    out = {}
    for k, v in model.decoder.state_dict().items():
        out["decoder." + k] = v.detach().cpu().numpy()
    for k, v in model.encoder.state_dict().items():
        out["encoder." + k] = v.detach().cpu().numpy()
    np.savez_compressed(REPO / "di_fusion_amd" / "network" / "weights_default.npz", **out)
    print("weights:", len(out), "tensors")


def golden_networks(model):
    g = torch.Generator().manual_seed(7)
    n = 384
    lat = torch.randn((n, 29), generator=g) * 0.3
    xyz = torch.rand((n, 3), generator=g) * 2.0 - 1.0
    x = torch.cat([lat, xyz], 1)
    with torch.no_grad():
        sdf, std = model.decoder(x)
    pts = torch.cat([torch.rand((n, 3), generator=g) * 2 - 1,
                     torch.nn.functional.normalize(torch.randn((n, 3), generator=g), dim=1)], 1)
    with torch.no_grad():
        enc = model.encoder(pts)
    lattices = {}
    for r in (2, 4, 8):
        a = -(r // 2) * (1. / r)
        b = 1. + (r - 1) // 2 * (1. / r)
        lattices[f"lattice_r{r}_default"] = ref_util.get_samples(r, torch.device("cpu")).numpy()
        lattices[f"lattice_r{r}_ab"] = ref_util.get_samples(r, torch.device("cpu"), a=a, b=b).numpy()
    # the lattices extract_mesh(resolution=4) really uses (map.py:640-646): R=8 and l=4 over [a,b], minus 0.5
    a, b = -(4 // 2) * (1. / 4), 1. + (4 - 1) // 2 * (1. / 4)
    lattices["extract_low_l4"] = (ref_util.get_samples(4, torch.device("cpu"), a=a, b=b) - 0.5).numpy()
    lattices["extract_high_R8"] = (ref_util.get_samples(8, torch.device("cpu"), a=a, b=b) - 0.5).numpy()
    # trilinear known answer (map.py:658-663)
    low = torch.randn((5, 1, 4, 4, 4), generator=g)
    up = torch.nn.functional.interpolate(low, mode="trilinear", size=(8, 8, 8), align_corners=True)
    np.savez_compressed(HERE / "networks.npz", dec_x=x.numpy(), dec_sdf=sdf.numpy(), dec_std=std.numpy(),
                        enc_x=pts.numpy(), enc_out=enc.numpy(), tri_low=low.numpy(), tri_up=up.numpy(), **lattices)
    print("networks: decoder/encoder/lattices/trilinear saved")


def map_state(m):
    n = int(m.n_occupied)
    idx = m.indexer.numpy()
    nz = np.nonzero(idx != -1)[0]
    return dict(n_occupied=np.int64(n), indexer_nz=nz.astype(np.int64), indexer_val=idx[nz].astype(np.int64),
                latent_vecs_pos=m.latent_vecs_pos[:n].numpy().copy(),
                voxel_obs_count=m.voxel_obs_count[:n].numpy().copy(),
                latent_vecs=m.latent_vecs[:n].numpy().copy(),
                capacity=np.int64(m.latent_vecs.size(0)),
                updated_vec_id=m.mesh_cache.updated_vec_id.numpy().copy())


def run_sequence(model, name, scene, cfg, intr, n_frames, deg_per_frame, store_inputs, store_cubes,
                 probe=True, extract_every=1):
    args = cfg.namespace()
    m = ref_map.DenseIndexedMap(model, args, 29, torch.device("cpu"))
    out = dict(n_frames=np.int64(n_frames), n_xyz=np.asarray(m.n_xyz, dtype=np.int64),
               bound_min=np.asarray(cfg.bound_min, dtype=np.float64), voxel_size=np.float64(cfg.voxel_size),
               deg_per_frame=np.float64(deg_per_frame),
               intr=np.asarray([intr.fx, intr.fy, intr.cx, intr.cy, intr.width, intr.height], dtype=np.float64))
    for f in range(n_frames):
        xyz, nrm = syn.frame_points(scene, f, intr, deg_per_frame=deg_per_frame)
        out[f"f{f}_xyz_sha"] = np.asarray(sha(xyz.numpy()))
        out[f"f{f}_nrm_sha"] = np.asarray(sha(nrm.numpy()))
        out[f"f{f}_n_points"] = np.int64(xyz.size(0))
        if store_inputs:
            out[f"f{f}_xyz"] = xyz.numpy()
            out[f"f{f}_nrm"] = nrm.numpy()
        unq = m.integrate_keyframe(xyz, nrm)
        out[f"f{f}_unq_mask"] = np.packbits(unq.numpy())
        for k, v in map_state(m).items():
            out[f"f{f}_int_{k}"] = v
        if (f % extract_every) == 0:
            RECORDED.clear()
            _extract(m)
            a = RECORDED["mc_args"]
            out[f"f{f}_mc_valid_blocks"] = a["valid_blocks"].numpy()
            out[f"f{f}_mc_vec_batch_mapping"] = a["vec_batch_mapping"].numpy()
            out[f"f{f}_mc_B"] = np.int64(a["cube_sdf"].size(0))
            out[f"f{f}_mc_cube_sdf_sha"] = np.asarray(sha(a["cube_sdf"].numpy()))
            if store_cubes:
                out[f"f{f}_mc_cube_sdf"] = a["cube_sdf"].numpy()
                out[f"f{f}_mc_cube_std"] = a["cube_std"].numpy()
            else:   # keep a strided subset of voxels so that large cases still pin values
                sel = np.arange(0, a["cube_sdf"].size(0), max(1, a["cube_sdf"].size(0) // 64))
                out[f"f{f}_mc_cube_sel"] = sel.astype(np.int64)
                out[f"f{f}_mc_cube_sdf"] = a["cube_sdf"].numpy()[sel]
                out[f"f{f}_mc_cube_std"] = a["cube_std"].numpy()[sel]
        print(f"  {name} frame {f}: N={xyz.size(0)} kept={int(unq.sum())} n_occ={int(m.n_occupied)} "
              f"B={int(out.get(f'f{f}_mc_B', -1))}")
    if probe:
        g = torch.Generator().manual_seed(11)
        xyz, _ = syn.frame_points(scene, 0, intr, deg_per_frame=deg_per_frame)
        sel = torch.randperm(xyz.size(0), generator=g)[:512]
        q = xyz[sel] + (torch.rand((512, 3), generator=g) - 0.5) * cfg.voxel_size * 0.6
        with torch.no_grad():
            sdf, std, mask = m.get_sdf(q)
        out["probe_xyz"] = q.numpy()
        out["probe_sdf"] = sdf.numpy()
        out["probe_std"] = std.numpy()
        out["probe_mask"] = mask.numpy()
        out["probe_grad"] = reference_probe_gradient(m, q, sdf, mask)
    np.savez_compressed(HERE / f"{name}.npz", **out)
    print(f"{name}: saved ({(HERE / f'{name}.npz').stat().st_size / 1e6:.2f} MB)")


def reference_probe_gradient(m, q, sdf_no_grad, mask_no_grad):
    """what the tracker differentiates (tracker.py:184-192): `get_sdf(xyz.requires_grad_())` (map.py:559-579), residual
    sdf / std.detach(), `autograd.grad(residual, xyz, ones)`; rows of the valid points, (M, 3) float32."""
    qg = q.clone().requires_grad_(True)
    sdf, std, mask = m.get_sdf(qg)
    assert torch.equal(mask, mask_no_grad) and torch.allclose(sdf.detach(), sdf_no_grad, atol=0, rtol=0)
    res = sdf / std.detach()
    (g,) = torch.autograd.grad(res, [qg], grad_outputs=torch.ones_like(res), retain_graph=False, create_graph=False)
    return g[mask].numpy()


def add_gradient_probes(model, name, scene, cfg, intr, n_frames, deg_per_frame):
    """Add `probe_grad` to an existing sequence fixture without touching its other arrays: the map is rebuilt by the reference, the stored
    probe values must come out again bit for bit, then the reference's autograd gradient at the stored probe points is appended."""
    path = HERE / f"{name}.npz"
    old = {k: v for k, v in np.load(path).items()}
    m = ref_map.DenseIndexedMap(model, cfg.namespace(), 29, torch.device("cpu"))
    for f in range(n_frames):
        xyz, nrm = syn.frame_points(scene, f, intr, deg_per_frame=deg_per_frame)
        assert sha(xyz.numpy()) == str(old[f"f{f}_xyz_sha"])
        m.integrate_keyframe(xyz, nrm)
        RECORDED.clear()
        _extract(m)
    assert np.array_equal(m.latent_vecs[:int(m.n_occupied)].numpy(), old[f"f{n_frames - 1}_int_latent_vecs"]), "the rebuilt map differs from the fixture"
    q = torch.from_numpy(old["probe_xyz"])
    with torch.no_grad():
        sdf, std, mask = m.get_sdf(q)
    assert np.array_equal(sdf.numpy(), old["probe_sdf"]) and np.array_equal(mask.numpy(), old["probe_mask"])
    old["probe_grad"] = reference_probe_gradient(m, q, sdf, mask)
    np.savez_compressed(path, **old)
    print(f"{name}: probe_grad {old['probe_grad'].shape} added, |g| max {np.abs(old['probe_grad']).max():.3f}")


def _extract(m):
    """`extract_mesh` up to the marching-cubes call; `_make_mesh_from_cache` needs open3d so the mesh-building
    tail is cut off by making it a no-op (it runs after everything we record)."""
    m._make_mesh_from_cache = lambda: None
    return m.extract_mesh(4, int(4e6), max_std=0.15, extract_async=False, interpolate=True)


def save_reference_map(model):
    """A map written by the reference's own `DenseIndexedMap.save` (map.py:239-243) after the three seq_small frames:
    the wire format `di_fusion_amd`'s `load` must keep opening."""
    intr_s = syn.Intrinsic().scaled(0.125)
    cfg = syn.MapConfig((-1.6, -1.6, -1.6), (1.6, 1.6, 1.6), 0.4)
    m = ref_map.DenseIndexedMap(model, cfg.namespace(), 29, torch.device("cpu"))
    for f in range(3):
        xyz, nrm = syn.frame_points(syn.Scene(kind="sphere", radius=1.3), f, intr_s, deg_per_frame=20.0)
        m.integrate_keyframe(xyz, nrm)
    m.save(HERE / "ref_map_small.pt")
    print("ref_map_small.pt:", (HERE / "ref_map_small.pt").stat().st_size, "bytes, keys", sorted(m.cold_vars.keys()))


def optimize_sequence(model):
    """The latent-optimisation stage (map.py:459-513, `integrate_keyframe(do_optimize=True)`, synchronous): seq_small's frames with
    optim_n_iters = 5 and the code regulariser on.  The reference draws its sample perturbations from torch.randn; every draw is
    recorded (in call order), so that a restatement can be fed the same numbers."""
    intr = syn.Intrinsic().scaled(0.125)
    scene = syn.Scene(kind="sphere", radius=1.3)
    cfg = syn.MapConfig((-1.6, -1.6, -1.6), (1.6, 1.6, 1.6), 0.4)
    args = cfg.namespace()
    args.optim_n_iters, args.code_regularization, args.code_reg_lambda = 5, True, 1.0e-2
    m = ref_map.DenseIndexedMap(model, args, 29, torch.device("cpu"))
    out = dict(n_frames=np.int64(4), optim_n_iters=np.int64(5), code_reg_lambda=np.float64(1.0e-2), deg_per_frame=np.float64(20.0))
    draws = []
    orig_randn = torch.randn

    def recording_randn(*a, **k):
        t = orig_randn(*a, **k)
        draws.append(t.clone())
        return t

    # every loss term the reference's optimiser evaluates (map.py:86-101: "ll" before each Adam step, "reg" when the regulariser is on)
    from utils import exp_util
    losses = []
    orig_add = exp_util.CombinedChunkLoss.add_loss

    def recording_add(self, name, val):
        losses.append((name, float(val.item())))
        return orig_add(self, name, val)

    exp_util.CombinedChunkLoss.add_loss = recording_add
    torch.manual_seed(4321)
    for f in range(4):
        xyz, nrm = syn.frame_points(scene, f, intr, deg_per_frame=20.0)
        draws.clear()
        losses.clear()
        torch.randn = recording_randn
        try:
            m.integrate_keyframe(xyz, nrm, do_optimize=True, async_optimize=False)
        finally:
            torch.randn = orig_randn
        out[f"f{f}_noise"] = torch.cat(draws).numpy() if draws else np.zeros((0,), np.float32)
        out[f"f{f}_loss_ll"] = np.asarray([v for k, v in losses if k == "ll"], np.float64)       # one per Adam iteration (a single chunk here)
        out[f"f{f}_loss_reg"] = np.asarray([v for k, v in losses if k == "reg"], np.float64)
        n = int(m.n_occupied)
        out[f"f{f}_latent_vecs"] = m.latent_vecs[:n].numpy().copy()
        out[f"f{f}_voxel_obs_count"] = m.voxel_obs_count[:n].numpy().copy()
        out[f"f{f}_voxel_optimized"] = m.voxel_optimized[:n].numpy().copy()
        out[f"f{f}_updated_vec_id"] = m.mesh_cache.updated_vec_id.numpy().copy()
        RECORDED.clear()
        _extract(m)
        print(f"  seq_optim frame {f}: n_occ={n} optimised so far={int(m.voxel_optimized[:n].sum())} noise draws={out[f'f{f}_noise'].shape[0]}")
    exp_util.CombinedChunkLoss.add_loss = orig_add
    if (HERE / "seq_optim.npz").exists():
        # Everything that was in the fixture before must come out again: integer state bit for bit, floats to the reference's own
        # run-to-run noise (its CPU index_add_ / BLAS sums are not bit-reproducible between runs: latents move by ~1e-6 between two runs).  The
        # arrays already in the fixture are KEPT (tests were validated against them); only new keys are added.
        old = np.load(HERE / "seq_optim.npz")
        for k in old.files:
            a, b = old[k], out[k]
            assert a.shape == b.shape and a.dtype == b.dtype, f"seq_optim: {k} changed shape"
            if a.dtype.kind == "f" and k.endswith("latent_vecs"):
                assert np.abs(a - b).max() <= 1e-5, f"seq_optim: {k} moved by {np.abs(a - b).max()}"
            else:
                assert np.array_equal(a, b), f"seq_optim: {k} changed"
            out[k] = a
    np.savez_compressed(HERE / "seq_optim.npz", **out)
    print(f"seq_optim: saved ({(HERE / 'seq_optim.npz').stat().st_size / 1e6:.2f} MB)")


def full_size_sequences(model, which):
    """BASELINE configs C2 (64^3) and C3 (128^3) at their real size: two full 640x480 frames of the bench stream (room scene,
    0.5 deg per frame), whole map state after every integrate, a strided subset of the decoded cubes."""
    if "seq_c2" in which:
        scene, cfg = syn.config_c2()
        run_sequence(model, "seq_c2", scene, cfg, syn.Intrinsic(), n_frames=2, deg_per_frame=0.5,
                     store_inputs=False, store_cubes=False)
    if "seq_c3" in which:
        scene, cfg = syn.config_c3()
        run_sequence(model, "seq_c3", scene, cfg, syn.Intrinsic(), n_frames=2, deg_per_frame=0.5,
                     store_inputs=False, store_cubes=False)


def long_sequence(model, name, scene, cfg, intr, n_frames, deg_per_frame, phase_deg=0.0, n_lat=256, n_cubes=16):
    """the reference stepped over the WINDOW the bench is timed on (frames 0..24 of the C3 stream: the driver's
    `--warmup 5 --steps 20`), and a C2 run long enough for the 600-count gate (map.py:409-410) to freeze a large share of the voxels and
    for cached triangles to be replaced (map.py:703-714).  Per frame the whole integer state as SHA-256 of the arrays `check_state`
    compares elsewhere (so the fixture stays small), the sizes, a strided subset of latents and decoded cubes, and — last frame — all
    latents.  Consumed frame by frame through the MEASURED entry points (`FusionStream.step_direct(d2h="dma")`, `FusionStreamGroup`)
    by tests/test_gpu_long.py and by the oracle in tests/test_oracle_golden.py."""
    args = cfg.namespace()
    m = ref_map.DenseIndexedMap(model, args, 29, torch.device("cpu"))
    out = dict(n_frames=np.int64(n_frames), n_xyz=np.asarray(m.n_xyz, dtype=np.int64), deg_per_frame=np.float64(deg_per_frame),
               phase_deg=np.float64(phase_deg), encoder_count_th=np.float64(args.encoder_count_th))
    import time
    for f in range(n_frames):
        t0 = time.time()
        xyz, nrm = syn.frame_points(scene, f, intr, deg_per_frame=deg_per_frame, phase_deg=phase_deg)
        out[f"f{f}_xyz_sha"] = np.asarray(sha(xyz.numpy()))
        out[f"f{f}_n_points"] = np.int64(xyz.size(0))
        n_before = int(m.n_occupied)
        gated_before = int((m.voxel_obs_count[:n_before] > args.encoder_count_th).sum())
        unq = m.integrate_keyframe(xyz, nrm)
        st = map_state(m)
        n = int(st["n_occupied"])
        out[f"f{f}_n_occupied"] = np.int64(n)
        out[f"f{f}_capacity"] = st["capacity"]
        out[f"f{f}_n_gated_before"] = np.int64(gated_before)          # voxels the encoder no longer updates in this frame
        out[f"f{f}_unq_mask_sha"] = np.asarray(sha(np.packbits(unq.numpy())))
        out[f"f{f}_n_kept"] = np.int64(int(unq.sum()))
        for k in ("indexer_nz", "indexer_val", "latent_vecs_pos", "voxel_obs_count", "updated_vec_id"):
            out[f"f{f}_{k}_sha"] = np.asarray(sha(st[k]))
        out[f"f{f}_n_updated"] = np.int64(st["updated_vec_id"].shape[0])
        sel = np.unique(np.linspace(0, n - 1, n_lat).astype(np.int64))
        out[f"f{f}_lat_sel"] = sel
        out[f"f{f}_lat"] = st["latent_vecs"][sel]
        if f == n_frames - 1:
            out["last_latent_vecs"] = st["latent_vecs"]
            out["last_voxel_obs_count"] = st["voxel_obs_count"]
            out["last_latent_vecs_pos"] = st["latent_vecs_pos"]
        RECORDED.clear()
        _extract(m)
        a = RECORDED["mc_args"]
        vb = a["valid_blocks"].numpy()
        vbm = a["vec_batch_mapping"].numpy()
        B = int(a["cube_sdf"].size(0))
        occ = np.full((B,), -1, np.int64)
        nzv = np.nonzero(vbm >= 0)[0]
        occ[vbm[nzv]] = nzv                                             # batch row -> slot (= occupied_vec_id, map.py:633-637)
        assert (occ >= 0).all()
        out[f"f{f}_mc_valid_blocks_sha"] = np.asarray(sha(vb.astype(np.int64)))
        out[f"f{f}_mc_K"] = np.int64(vb.shape[0])
        out[f"f{f}_mc_occ_sha"] = np.asarray(sha(occ))
        out[f"f{f}_mc_B"] = np.int64(B)
        csel = np.unique(np.linspace(0, B - 1, n_cubes).astype(np.int64))
        out[f"f{f}_mc_cube_sel"] = csel
        out[f"f{f}_mc_cube_sdf"] = a["cube_sdf"].numpy()[csel]
        out[f"f{f}_mc_cube_std"] = a["cube_std"].numpy()[csel]
        print(f"  {name} frame {f}: N={xyz.size(0)} kept={int(unq.sum())} n_occ={n} gated_before={gated_before} "
              f"updated={st['updated_vec_id'].shape[0]} K={vb.shape[0]} B={B}  ({time.time() - t0:.1f} s)", flush=True)
    np.savez_compressed(HERE / f"{name}.npz", **out)
    print(f"{name}: saved ({(HERE / f'{name}.npz').stat().st_size / 1e6:.2f} MB)")


def long_sequences(model, which):
    if "seq_c3_long" in which:
        scene, cfg = syn.config_c3()
        long_sequence(model, "seq_c3_long", scene, cfg, syn.Intrinsic(), n_frames=25, deg_per_frame=0.5)
    if "seq_c3_long_p45" in which:          # the second stream of a group: another arc of the same orbit
        scene, cfg = syn.config_c3()
        long_sequence(model, "seq_c3_long_p45", scene, cfg, syn.Intrinsic(), n_frames=12, deg_per_frame=0.5, phase_deg=45.0, n_cubes=8)
    if "seq_c2_long" in which:
        scene, cfg = syn.config_c2()
        long_sequence(model, "seq_c2_long", scene, cfg, syn.Intrinsic(), n_frames=16, deg_per_frame=0.5)


# ---- the tracker's SDF term (tracker.py:174-283) ---------------------------------------------------------------------
def _import_reference_tracker():
    pq = types.ModuleType("pyquaternion")

    class Quaternion:
        """Rotation kept as a matrix (see the module docstring)."""

        def __init__(self, *a, matrix=None, degrees=None, axis=None, **k):
            assert not a and not k
            if matrix is not None:
                u, _, vt = np.linalg.svd(np.asarray(matrix, dtype=np.float64)[:3, :3])       # the nearest rotation, as a unit quaternion would be
                self.m = u @ vt
            elif degrees is not None:
                ax = np.asarray(axis, dtype=np.float64)
                ax = ax / np.linalg.norm(ax)
                th = np.radians(degrees)
                K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
                self.m = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
            else:
                self.m = np.eye(3)

        rotation_matrix = property(lambda self: self.m)
        inverse = property(lambda self: Quaternion(matrix=self.m.T))

        def rotate(self, v):
            return self.m @ np.asarray(v)

        def __mul__(self, other):
            return Quaternion(matrix=self.m @ other.m)

    pq.Quaternion = Quaternion
    sys.modules["pyquaternion"] = pq
    for name in ("unproject_depth", "remove_radius_outlier", "estimate_normals", "rgb_odometry", "gradient_xy"):
        setattr(_ext, name, None)
    torch.Tensor.cuda = lambda self, *a, **k: self
    from system import tracker as ref_tracker
    from utils.motion_util import Isometry
    return ref_tracker, Isometry, Quaternion


def tracker_fixture(model, name, scene, cfg, intr_map, intr_obs, n_map_frames, deg_per_frame, cases, iter_config, init_deg=None, obs_frame=None,
                    obs_phase_deg=0.0):
    """The reference's map after `n_map_frames` frames, then its tracker's SDF term for frame `n_map_frames` of the same orbit: the numbers
    `SDFTracker.compute_sdf_Hg` returns for each (delta pose, robust kernel) of `cases`, and `SDFTracker.gauss_newton` over `iter_config`
    ('sdf' terms only) from the previous frame's pose — every call it makes to compute_sdf_Hg recorded."""
    import argparse
    ref_tracker, Isometry, Quaternion = _import_reference_tracker()
    m = ref_map.DenseIndexedMap(model, cfg.namespace(), 29, torch.device("cpu"))
    out = dict(n_map_frames=np.int64(n_map_frames), deg_per_frame=np.float64(deg_per_frame))
    for f in range(n_map_frames):
        xyz, nrm = syn.frame_points(scene, f, intr_map, deg_per_frame=deg_per_frame)
        out[f"f{f}_xyz_sha"] = sha(xyz.numpy())
        m.integrate_keyframe(xyz, nrm)
    out["n_occupied"] = np.int64(int(m.n_occupied))
    out["latent_vecs"] = m.latent_vecs[:int(m.n_occupied)].numpy().copy()
    out["latent_pos"] = m.latent_vecs_pos[:int(m.n_occupied)].numpy().copy()
    out["voxel_obs_count"] = m.voxel_obs_count[:int(m.n_occupied)].numpy().copy()
    obs_frame = n_map_frames if obs_frame is None else obs_frame
    out["obs_frame"], out["obs_phase_deg"] = np.int64(obs_frame), np.float64(obs_phase_deg)
    obs, R_gt, t_gt = syn.frame_cloud_camera(scene, obs_frame, intr_obs, deg_per_frame=deg_per_frame, phase_deg=obs_phase_deg)
    R_last, t_last = syn.orbit_pose(n_map_frames - 1, deg_per_frame=deg_per_frame)
    out["obs_sha"] = sha(obs.numpy())
    out["obs_n"] = np.int64(obs.size(0))
    out["last_R"], out["last_t"], out["gt_R"], out["gt_t"] = R_last, t_last, R_gt, t_gt
    last_pose = Isometry(q=Quaternion(matrix=R_last), t=t_last)

    def tracker(kernel, k, iters):
        a = argparse.Namespace(sdf=dict(robust_kernel=kernel, robust_k=k, subsample=0.5),
                               rgb=dict(weight=500.0, robust_kernel=None, robust_k=0.01, min_grad_scale=0.0, max_depth_delta=0.2), iter_config=iters)
        t = ref_tracker.SDFTracker(m, a)
        t.all_pd_pose = [last_pose]
        return t

    for i, (xi, kernel, k) in enumerate(cases):
        delta = Isometry.from_twist(np.asarray(xi, dtype=np.float64))
        t = tracker(kernel, k, [])
        H, g, e = t.compute_sdf_Hg(0, last_pose, delta, obs.clone(), False)
        _, _, e2 = t.compute_sdf_Hg(-1, last_pose, delta, obs.clone(), True)
        assert e2 == e
        cur = (last_pose.dot(delta)) @ obs
        with torch.no_grad():
            _, _, mask = m.get_sdf(cur)
        out[f"case{i}_xi"] = np.asarray(xi, dtype=np.float64)
        out[f"case{i}_kernel"] = np.array(str(kernel))
        out[f"case{i}_k"] = np.float64(k)
        out[f"case{i}_delta_R"], out[f"case{i}_delta_t"] = delta.q.rotation_matrix.copy(), delta.t.copy()
        out[f"case{i}_H"], out[f"case{i}_g"], out[f"case{i}_e"], out[f"case{i}_M"] = H, g, np.float64(e), np.int64(int(mask.sum()))
        out[f"case{i}_mask_sha"] = sha(mask.numpy())
        print(f"{name} case {i}: kernel {kernel} k {k} M {int(mask.sum())} / {obs.size(0)} e {e:.6f} |H| {np.abs(H).max():.4f} |g| {np.abs(g).max():.5f}")

    # the loop: from the previous frame's pose (tracker.py:121-122: `lspeed` = identity), or from a pose `init_deg` of yaw further off
    t = tracker("huber", 5.0, iter_config)
    calls = []
    orig = t.compute_sdf_Hg

    def recording(n_iter, last_pose_, cur_delta_pose, obs_xyz, no_grad=False):
        r = orig(n_iter, last_pose_, cur_delta_pose, obs_xyz, no_grad)
        calls.append((n_iter, cur_delta_pose.q.rotation_matrix.copy(), cur_delta_pose.t.copy(), r))
        return r

    t.compute_sdf_Hg = recording
    init = last_pose
    if init_deg is not None:
        th = np.radians(init_deg)
        init = last_pose.dot(Isometry(q=Quaternion(matrix=np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]]))))
    out["gn_init_R"], out["gn_init_t"] = init.q.rotation_matrix.copy(), init.t.copy()
    final = t.gauss_newton(init, None, None, None, obs.clone(), None)
    out["gn_iter_config"] = np.array(json.dumps(iter_config))
    out["gn_n_calls"] = np.int64(len(calls))
    for j, (n_iter, dR, dt, (H, g, e)) in enumerate(calls):
        out[f"gn{j}_iter"], out[f"gn{j}_delta_R"], out[f"gn{j}_delta_t"], out[f"gn{j}_e"] = np.int64(n_iter), dR, dt, np.float64(e)
        if H is not None:
            out[f"gn{j}_H"], out[f"gn{j}_g"] = H, g
    out["gn_final_R"], out["gn_final_t"] = final.q.rotation_matrix.copy(), final.t.copy()
    err_t = np.linalg.norm(final.t - t_gt)
    err_R = np.degrees(np.arccos(np.clip((np.trace(final.q.rotation_matrix.T @ R_gt) - 1) / 2, -1, 1)))
    e0 = calls[0][3][2]
    print(f"{name} gauss_newton: {len(calls)} calls, energy {e0:.6f} -> {calls[-1][3][2]:.6f}, final pose off the true one by {err_t * 1000:.2f} mm / {err_R:.4f} deg")
    np.savez_compressed(HERE / f"{name}.npz", **out)
    print(f"{name}: saved ({(HERE / f'{name}.npz').stat().st_size / 1e6:.2f} MB)")


def box_filter_fixture():
    """The reference's own `point_box_filter` (system/tracker.py:13-23, plain torch) on the tracker's half-resolution cloud of the room stream —
    with `torch_scatter.scatter_mean` (not in this image) as an index_add_ / count stand-in: sums in float64, then float32 like scatter_mean's
    result.  Inputs by hash (synthetic frame 3 at 320 x 240), outputs stored."""
    ref_tracker, _, _ = _import_reference_tracker()
    ts = types.ModuleType("torch_scatter")

    def scatter_mean(src, index, dim=0):
        n = int(index.max()) + 1
        out = torch.zeros((n, src.size(1)), dtype=torch.float64)
        out.index_add_(0, index, src.double())
        cnt = torch.zeros((n,), dtype=torch.float64)
        cnt.index_add_(0, index, torch.ones_like(index, dtype=torch.float64))
        return (out / cnt[:, None]).float()

    ts.scatter_mean = scatter_mean
    sys.modules["torch_scatter"] = ts
    xyz, nrm = syn.frame_points(syn.default_room(), 3, syn.Intrinsic().scaled(0.5))
    out = dict(xyz_sha=np.array(sha(xyz.numpy())), nrm_sha=np.array(sha(nrm.numpy())), n=np.int64(xyz.size(0)))
    for vs in (0.02, 0.05):
        fp, fn = ref_tracker.point_box_filter(xyz, nrm, vs)
        out[f"vs{int(vs * 100)}_points"], out[f"vs{int(vs * 100)}_normals"] = fp.numpy(), fn.numpy()
        print(f"box_filter: voxel {vs}: {xyz.size(0)} points -> {fp.size(0)} boxes")
    np.savez_compressed(HERE / "box_filter.npz", **out)
    print(f"box_filter: saved ({(HERE / 'box_filter.npz').stat().st_size / 1e6:.2f} MB)")


def tracker_fixtures(model):
    cases = [((0, 0, 0, 0, 0, 0), "huber", 5.0),
             ((0.01, -0.004, 0.006, 0.004, -0.008, 0.003), "huber", 5.0),
             ((0.01, -0.004, 0.006, 0.004, -0.008, 0.003), "huber", 0.5),
             ((-0.006, 0.003, 0.002, -0.002, 0.005, 0.001), "tukey", 1.5),
             ((0.002, 0.001, -0.003, 0.001, 0.002, -0.001), None, 0.0)]
    iters = [{"n": 10, "type": [["sdf"]]}, {"n": 5, "type": [["sdf"]]}]
    # S: seq_room16's scene and map (16^3 grid of 0.4 m voxels, four 160 x 120 frames 15 degrees apart), a 160 x 120 cloud seen one degree
    # further along the orbit than the last frame
    tracker_fixture(model, "track_small", syn.default_room(), syn.MapConfig((-3.2, -3.2, -3.2), (3.2, 3.2, 3.2), 0.4),
                    syn.Intrinsic().scaled(0.25), syn.Intrinsic().scaled(0.25), 4, 15.0, cases, iters, init_deg=None, obs_frame=3, obs_phase_deg=1.0)
    # C2 of BASELINE.json (64^3 grid, 0.1 m voxels, room scene), two 640x480 frames in the map, the tracker's 320x240 cloud of frame 2
    scene, cfg = syn.config_c2()
    tracker_fixture(model, "track_c2", scene, cfg, syn.Intrinsic(), syn.Intrinsic().scaled(0.5), 2, 0.5, cases, iters, init_deg=0.4)


VARIANTS = [("accumulated", dict(voxel_resolution=4, fast=True, no_cache=False)),      # two integrates, then ONE extract: the dirty sets accumulate (map.py:303-308)
            ("exact_r4", dict(voxel_resolution=4, fast=False, no_cache=True)),           # every lattice sample decoded (map.py:683-685)
            ("fast_r2", dict(voxel_resolution=2, fast=True, no_cache=True)),
            ("fast_r3", dict(voxel_resolution=3, fast=True, no_cache=True)),
            ("fast_r8", dict(voxel_resolution=8, fast=True, no_cache=True)),
            ("exact_r2", dict(voxel_resolution=2, fast=False, no_cache=True)),
            ("no_cache_r4", dict(voxel_resolution=4, fast=True, no_cache=True))]         # every allocated voxel re-meshed (map.py:614-616)


def variants_fixture(model):
    """`extract_mesh` AWAY from the call the tracking loop makes (resolution 4, fast, cached) and `integrate_keyframe` without pruning, run by the
    REFERENCE on CPU: the 16^3 room map after two integrates WITHOUT an extract in between, then one extract per entry of VARIANTS (the argument
    tuple handed to marching cubes: dirty list, batch map, B, a strided subset of the cubes), and a second map with `prune_min_vox_obs = 0`
    (map.py:372-378: no mask is returned, every point's voxel is allocated).  -> extract_variants.npz"""
    scene, cfg, intr = syn.default_room(), syn.MapConfig((-3.2, -3.2, -3.2), (3.2, 3.2, 3.2), 0.4), syn.Intrinsic().scaled(0.25)
    deg = 15.0
    out = dict(deg_per_frame=np.float64(deg), n_frames=np.int64(2), names=np.asarray([n for n, _ in VARIANTS]))
    m = ref_map.DenseIndexedMap(model, cfg.namespace(), 29, torch.device("cpu"))
    for f in range(2):
        xyz, nrm = syn.frame_points(scene, f, intr, deg_per_frame=deg)
        out[f"f{f}_xyz_sha"] = np.asarray(sha(xyz.numpy()))
        m.integrate_keyframe(xyz, nrm)
    for k, v in map_state(m).items():
        out[f"int_{k}"] = v
    for name, kw in VARIANTS:
        RECORDED.clear()
        m._make_mesh_from_cache = lambda: None
        m.extract_mesh(kw["voxel_resolution"], int(4e6), fast=kw["fast"], max_std=0.15, extract_async=False, no_cache=kw["no_cache"], interpolate=True)
        a = RECORDED["mc_args"]
        B = a["cube_sdf"].size(0)
        sel = np.arange(0, B, max(1, B // 24))
        out[f"{name}_resolution"] = np.int64(kw["voxel_resolution"]); out[f"{name}_fast"] = np.int64(kw["fast"]); out[f"{name}_no_cache"] = np.int64(kw["no_cache"])
        out[f"{name}_valid_blocks"] = a["valid_blocks"].numpy()
        out[f"{name}_vec_batch_mapping"] = a["vec_batch_mapping"].numpy()
        out[f"{name}_B"] = np.int64(B)
        out[f"{name}_cube_sdf_sha"] = np.asarray(sha(a["cube_sdf"].numpy()))
        out[f"{name}_cube_sel"] = sel.astype(np.int64)
        out[f"{name}_cube_sdf"] = a["cube_sdf"].numpy()[sel]
        out[f"{name}_cube_std"] = a["cube_std"].numpy()[sel]
        out[f"{name}_updated_after"] = m.mesh_cache.updated_vec_id.numpy().copy()
        print(f"  variant {name}: K={a['valid_blocks'].numel()} B={B} cube {tuple(a['cube_sdf'].shape[1:])}")
    # pruning off
    args = cfg.namespace()
    args.prune_min_vox_obs = 0
    m2 = ref_map.DenseIndexedMap(model, args, 29, torch.device("cpu"))
    for f in range(2):
        xyz, nrm = syn.frame_points(scene, f, intr, deg_per_frame=deg)
        unq = m2.integrate_keyframe(xyz, nrm)
        assert unq is None
        for k, v in map_state(m2).items():
            out[f"noprune_f{f}_{k}"] = v
    RECORDED.clear()
    _extract(m2)
    a = RECORDED["mc_args"]
    out["noprune_valid_blocks"] = a["valid_blocks"].numpy()
    out["noprune_B"] = np.int64(a["cube_sdf"].size(0))
    sel = np.arange(0, a["cube_sdf"].size(0), max(1, a["cube_sdf"].size(0) // 24))
    out["noprune_cube_sel"] = sel.astype(np.int64)
    out["noprune_cube_sdf"] = a["cube_sdf"].numpy()[sel]
    out["noprune_cube_std"] = a["cube_std"].numpy()[sel]
    np.savez_compressed(HERE / "extract_variants.npz", **out)
    print(f"extract_variants: saved ({(HERE / 'extract_variants.npz').stat().st_size / 1e6:.2f} MB), no-prune map {int(m2.n_occupied)} voxels against {int(m.n_occupied)}")


def mesh_cache_fixture(model):
    """The reference's OWN code behind marching cubes (map.py:698-714: `vertices * voxel_size + bound_min`, `_get_valid_idx`, the replace-by-voxel
    concatenation of the mesh cache), run over four frames of the 16^3 room sequence.  The CUDA kernel cannot run here: for THIS fixture
    `system.ext.marching_cubes_interp` is the oracle's C restatement (oracle/mc_oracle.c) on the cubes the reference hands it, so what the vectors pin is
    the host-side rule, given triangles.  Stand-in stated: numba's `_get_valid_idx` (map.py:20-26) reads `query_idx[len]` unchecked when a cached id
    lies above every new id (garbage that is != v: the triangle is kept); the identity `jit` would raise there, so the function is handed `query_idx`
    with one int64-max sentinel appended (same answer, no out-of-range read).  -> mesh_cache.npz"""
    from oracle import difusion_oracle as O
    scene, cfg, intr = syn.default_room(), syn.MapConfig((-3.2, -3.2, -3.2), (3.2, 3.2, 3.2), 0.4), syn.Intrinsic().scaled(0.25)
    deg, F = 15.0, 4
    new = {}

    def mc(indexer, valid_blocks, vec_batch_mapping, cube_sdf, cube_std, max_n, n_xyz, max_std):
        t, i, d = O.marching_cubes_interp(indexer.numpy(), valid_blocks.numpy(), vec_batch_mapping.numpy(), cube_sdf.numpy(), cube_std.numpy(),
                                          int(max_n), list(n_xyz), float(max_std))
        new.update(tri=t, tid=i, tstd=d)
        return torch.from_numpy(t), torch.from_numpy(i), torch.from_numpy(d)

    real_valid = ref_map._get_valid_idx
    sentinel_used = [0]

    def valid_idx(base_idx, query_idx):
        sentinel_used[0] += int(base_idx.max() > query_idx.max())
        return real_valid(base_idx, np.concatenate([query_idx, np.asarray([np.iinfo(np.int64).max], dtype=query_idx.dtype)]))

    out = dict(deg_per_frame=np.float64(deg), n_frames=np.int64(F))
    _ext.marching_cubes_interp, ref_map._get_valid_idx = mc, valid_idx
    try:
        m = ref_map.DenseIndexedMap(model, cfg.namespace(), 29, torch.device("cpu"))
        m._make_mesh_from_cache = lambda: None
        for f in range(F):
            xyz, nrm = syn.frame_points(scene, f, intr, deg_per_frame=deg)
            out[f"f{f}_xyz_sha"] = np.asarray(sha(xyz.numpy()))
            m.integrate_keyframe(xyz, nrm)
            m.extract_mesh(4, int(4e6), fast=True, max_std=0.15, extract_async=False, no_cache=False, interpolate=True)
            c = m.mesh_cache
            out[f"f{f}_new_id"] = new["tid"].copy()
            out[f"f{f}_new_tri_voxel_units"] = new["tri"].copy()
            out[f"f{f}_new_std"] = new["tstd"].copy()
            out[f"f{f}_cache_id"] = c.vertices_flatten_id.copy()
            out[f"f{f}_cache_vertices_sha"] = np.asarray(sha(c.vertices))
            out[f"f{f}_cache_std_sha"] = np.asarray(sha(c.vertices_std))
            assert c.vertices.dtype == np.float32 and c.vertices_flatten_id.dtype == np.int64
            print(f"  mesh cache frame {f}: {new['tid'].shape[0]} new triangles over {len(np.unique(new['tid']))} voxels -> cache {c.vertices.shape[0]}")
        out["final_vertices"], out["final_std"] = c.vertices.copy(), c.vertices_std.copy()
    finally:
        _ext.marching_cubes_interp, ref_map._get_valid_idx = _mc_record, real_valid
    np.savez_compressed(HERE / "mesh_cache.npz", **out)
    print(f"mesh_cache: saved ({(HERE / 'mesh_cache.npz').stat().st_size / 1e6:.2f} MB); cached ids above every new id in {sentinel_used[0]} of {F - 1} updates")


def main():
    model, hyper = load_reference_model()
    if "--map-only" in sys.argv:
        save_reference_map(model)
        return
    if "--only" in sys.argv:
        which = sys.argv[sys.argv.index("--only") + 1].split(",")
        full_size_sequences(model, which)
        long_sequences(model, which)
        if "seq_optim" in which:
            optimize_sequence(model)
        if "track" in which:
            tracker_fixtures(model)
        if "box_filter" in which:
            box_filter_fixture()
        if "variants" in which:
            variants_fixture(model)
        if "mesh_cache" in which:
            mesh_cache_fixture(model)
        if "grads" in which:
            add_gradient_probes(model, "seq_small", syn.Scene(kind="sphere", radius=1.3), syn.MapConfig((-1.6, -1.6, -1.6), (1.6, 1.6, 1.6), 0.4),
                                syn.Intrinsic().scaled(0.125), 3, 20.0)
            add_gradient_probes(model, "seq_c2", *syn.config_c2(), syn.Intrinsic(), 2, 0.5)
        return
    export_weights(model)
    golden_networks(model)

    # S: tiny, inputs + full cubes stored.  80x60 image, 8^3 grid of 0.4 m voxels around a r=1.3 sphere.
    intr_s = syn.Intrinsic().scaled(0.125)
    run_sequence(model, "seq_small", syn.Scene(kind="sphere", radius=1.3),
                 syn.MapConfig((-1.6, -1.6, -1.6), (1.6, 1.6, 1.6), 0.4), intr_s,
                 n_frames=3, deg_per_frame=20.0, store_inputs=True, store_cubes=True)
    # M: 160x120 image, 16^3 grid, room scene with boxes, 4 frames with a large yaw so that new voxels,
    # re-allocation (capacity doubling), the 600-count gate and the running average are all exercised.
    intr_m = syn.Intrinsic().scaled(0.25)
    run_sequence(model, "seq_room16", syn.default_room(),
                 syn.MapConfig((-3.2, -3.2, -3.2), (3.2, 3.2, 3.2), 0.4), intr_m,
                 n_frames=4, deg_per_frame=15.0, store_inputs=False, store_cubes=False)
    # C1 of BASELINE.json: one full 640x480 frame, 32^3 grid (0.1 m), camera inside a 1.5 m sphere.
    scene, cfg = syn.config_c1()
    run_sequence(model, "seq_c1", scene, cfg, syn.Intrinsic(), n_frames=1, deg_per_frame=0.5,
                 store_inputs=False, store_cubes=False)
    save_reference_map(model)
    full_size_sequences(model, ("seq_c2", "seq_c3"))
    optimize_sequence(model)
    variants_fixture(model)
    mesh_cache_fixture(model)


if __name__ == "__main__":
    main()
