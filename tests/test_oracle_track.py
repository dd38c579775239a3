"""CPU: the oracle's restatement of the tracker's SDF term (oracle.compute_sdf_hg / gauss_newton_sdf, reference tracker.py:174-283) against
fixtures the reference's own `SDFTracker` produced (tests/golden/make_golden.py --only track): H, g and the energy of `compute_sdf_Hg` for
five (pose, robust kernel) cases, and every evaluation of a `gauss_newton` run, on a 16^3 map and on BASELINE's C2."""
import hashlib
import json

import numpy as np
import pytest

from di_fusion_amd import synthetic as syn
from oracle import difusion_oracle as O
from tests.conftest import GOLDEN

TRACK = {
    "track_small": (syn.default_room(), syn.MapConfig((-3.2,) * 3, (3.2,) * 3, 0.4), syn.Intrinsic().scaled(0.25), syn.Intrinsic().scaled(0.25)),
    "track_c2": (*syn.config_c2(), syn.Intrinsic(), syn.Intrinsic().scaled(0.5)),
}
# The reference sums float32 in torch's order and its latents come from another BLAS (<= 1.1e-6 apart, tests/test_oracle_golden.py); a point
# within an ulp of a voxel face may fall on the other side (M may differ by a few of 76,800).  Relative to the largest entry (measured:
# <= 4.2e-5 for H, <= 4.0e-5 for g, the valid sets identical):
REL_TOL = 2e-4


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def track_inputs(name):
    """(golden, config, frames for the map as (xyz, nrm) numpy pairs, obs cloud numpy): the synthetic stream, checked against the fixture's hashes."""
    scene, cfg, intr_map, intr_obs = TRACK[name]
    g = np.load(GOLDEN / f"{name}.npz")
    frames = []
    for f in range(int(g["n_map_frames"])):
        xyz, nrm = syn.frame_points(scene, f, intr_map, deg_per_frame=float(g["deg_per_frame"]))
        assert _sha(xyz.numpy()) == str(g[f"f{f}_xyz_sha"])
        frames.append((xyz, nrm))
    obs, R_gt, t_gt = syn.frame_cloud_camera(scene, int(g["obs_frame"]), intr_obs, deg_per_frame=float(g["deg_per_frame"]),
                                             phase_deg=float(g["obs_phase_deg"]))
    assert _sha(obs.numpy()) == str(g["obs_sha"]) and obs.size(0) == int(g["obs_n"])
    return g, cfg, frames, obs


def n_cases(g):
    return len([k for k in g.files if k.startswith("case") and k.endswith("_xi")])


def kernel_of(g, i):
    k = str(g[f"case{i}_kernel"])
    return None if k == "None" else k


def close(got, want, what, rel=REL_TOL):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-12)
    assert err < rel, f"{what}: {err:.3e} of the largest entry"
    return err


@pytest.mark.parametrize("name", ["track_small", "track_c2"])
def test_sdf_term_matches_the_references_tracker(name, oracle_net):
    g, cfg, frames, obs = track_inputs(name)
    m = O.OracleMap(oracle_net, cfg.bound_min, cfg.bound_max, cfg.voxel_size)
    for xyz, nrm in frames:
        m.integrate_keyframe(xyz.numpy(), nrm.numpy())
    assert m.n_occupied == int(g["n_occupied"]) and np.array_equal(m.latent_vecs_pos[:m.n_occupied], g["latent_pos"])
    obs = obs.numpy()
    for i in range(n_cases(g)):
        dR, dt = O.twist_exp(g[f"case{i}_xi"])
        assert np.abs(dR - g[f"case{i}_delta_R"]).max() < 1e-12 and np.abs(dt - g[f"case{i}_delta_t"]).max() < 1e-12       # Isometry.from_twist
        H, gg, e, M = O.compute_sdf_hg(m, obs, g["last_R"], g["last_t"], dR, dt, kernel_of(g, i), float(g[f"case{i}_k"]))
        assert abs(M - int(g[f"case{i}_M"])) <= 3
        close(H, g[f"case{i}_H"], f"{name} case {i} H")
        close(gg, g[f"case{i}_g"], f"{name} case {i} g", rel=REL_TOL * np.abs(g[f"case{i}_H"]).max() / np.abs(g[f"case{i}_g"]).max())
        assert abs(e - float(g[f"case{i}_e"])) < REL_TOL * max(1.0, abs(float(g[f"case{i}_e"])))
        _, _, e2, M2 = O.compute_sdf_hg(m, obs, g["last_R"], g["last_t"], dR, dt, kernel_of(g, i), float(g[f"case{i}_k"]), no_grad=True)
        assert e2 == e and M2 == M


@pytest.mark.parametrize("name", ["track_small", "track_c2"])
def test_gauss_newton_follows_the_references_iterations(name, oracle_net):
    """Every evaluation the reference's loop made (pose, H, g, energy), the accepted / rejected steps, and the pose it returned.
    track_small: the first step is rejected (0.4 m voxels); track_c2: nine evaluations, 5 mm / 0.011 degrees off the true pose."""
    g, cfg, frames, obs = track_inputs(name)
    m = O.OracleMap(oracle_net, cfg.bound_min, cfg.bound_max, cfg.voxel_size)
    for xyz, nrm in frames:
        m.integrate_keyframe(xyz.numpy(), nrm.numpy())
    (R, t), calls = O.gauss_newton_sdf(m, obs.numpy(), g["last_R"], g["last_t"], g["gn_init_R"], g["gn_init_t"], json.loads(str(g["gn_iter_config"])))
    assert len(calls) == int(g["gn_n_calls"])
    for j, (it, dR, dt, H, gg, e) in enumerate(calls):
        assert it == int(g[f"gn{j}_iter"])
        assert np.abs(dR - g[f"gn{j}_delta_R"]).max() < 1e-5 and np.abs(dt - g[f"gn{j}_delta_t"]).max() < 1e-5
        assert abs(e - float(g[f"gn{j}_e"])) < 1e-3 * max(1.0, float(g[f"gn{j}_e"]))
        if H is not None:
            close(H, g[f"gn{j}_H"], f"{name} evaluation {j} H", rel=2e-3)
    assert np.abs(R - g["gn_final_R"]).max() < 1e-5 and np.abs(t - g["gn_final_t"]).max() < 1e-5


def test_pose_matches_the_references_isometry():
    """`di_fusion_amd.system.tracker.Pose` (host arithmetic only: runs without a GPU) against what the reference's `Isometry` produced for the
    fixtures: `from_twist` for the five cases, and the loop's pose updates `from_twist(xi) . delta` re-derived from consecutive evaluations."""
    from di_fusion_amd.system.tracker import Pose
    g = np.load(GOLDEN / "track_c2.npz")
    for i in range(n_cases(g)):
        p = Pose.from_twist(g[f"case{i}_xi"])
        assert np.abs(p.R - g[f"case{i}_delta_R"]).max() < 1e-12 and np.abs(p.t - g[f"case{i}_delta_t"]).max() < 1e-12
        assert np.abs(p.R @ p.R.T - np.eye(3)).max() < 1e-12
        q = p.inv().dot(p)
        assert np.abs(q.R - np.eye(3)).max() < 1e-12 and np.abs(q.t).max() < 1e-12
    assert np.abs(Pose.from_twist(np.zeros(6)).matrix - np.eye(4)).max() == 0.0
    tiny = Pose.from_twist([1e-3, 0, 0, 1e-10, 0, 0])                    # the first-order branch (|phi| ~ 0)
    assert np.abs(tiny.R - np.eye(3)).max() < 2e-10 and abs(tiny.t[0] - 1e-3) < 1e-12
    # the loop: delta_{j+1} = from_twist(solve(H_j, -g_j)) . delta_j whenever evaluation j was accepted and had derivatives
    for j in range(int(g["gn_n_calls"]) - 1):
        if f"gn{j}_H" not in g or int(g[f"gn{j + 1}_iter"]) != int(g[f"gn{j}_iter"]) + 1:
            continue
        step = Pose.from_twist(np.linalg.solve(g[f"gn{j}_H"], -g[f"gn{j}_g"])).dot(Pose(g[f"gn{j}_delta_R"], g[f"gn{j}_delta_t"]))
        assert np.abs(step.R - g[f"gn{j + 1}_delta_R"]).max() < 1e-10 and np.abs(step.t - g[f"gn{j + 1}_delta_t"]).max() < 1e-10
    x = np.random.default_rng(0).standard_normal((5, 3))
    last = Pose(g["last_R"], g["last_t"])
    assert np.abs((last @ x) - (x @ last.R.T + last.t)).max() == 0.0
