"""Times DenseIndexedMap.get_sdf on the C3 room map: values only and values + analytic d sdf / d xyz (what one
Gauss-Newton iteration of the tracker asks for, reference tracker.py:174-218), and one whole iteration of the SDF term: the reference's op
sequence on torch tensors against the one-call `dif_sdf_hg` (host wall clock: every iteration ends with H and g on the host).
Usage: python tools/bench_query.py [--frames 30]"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from di_fusion_amd import synthetic as syn                      # noqa: E402
from di_fusion_amd.network import utility as net_util            # noqa: E402
from di_fusion_amd.stream import FusionStream                    # noqa: E402

DEC_FLOP_PER_ROW = 98816


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def wall(fn, reps=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def reference_iteration(m, obs, last, delta, k=5.0):
    """tracker.py:174-218 as the reference writes it (huber), on this map: get_sdf with autograd, the Jacobian and the sums as torch ops,
    H / g / energy to the host."""
    cur = (last.dot(delta)) @ obs
    cur.requires_grad_(True)
    sdf, std, mask = m.get_sdf(cur)
    s = sdf / std.detach()
    (d,) = torch.autograd.grad(s, [cur], grad_outputs=torch.ones_like(s), retain_graph=False, create_graph=False)
    d = d[mask]
    s = s.detach()
    c = (delta @ obs)[mask]
    Lt = torch.from_numpy(last.R.astype(np.float32).T).to(obs.device)
    Lai = torch.mm(d, Lt)
    J = torch.cat([Lai, torch.linalg.cross(c, Lai)], dim=-1)
    w = torch.ones_like(s)
    sa = torch.abs(s)
    big = sa > k
    w[big] = k / sa[big]
    wf = s * w
    JW = J * w.unsqueeze(1)
    scale = 1.0 / wf.size(0)
    e = (s * wf).sum().item() * scale
    H = torch.einsum("na,nb->nab", JW, J).sum(0) * scale
    g = (J * wf.unsqueeze(1)).sum(0) * scale
    return H.cpu().numpy().astype(float), g.cpu().numpy().astype(float), float(e)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=30)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    scene, cfg = syn.config_c3()
    model = net_util.networks_from_arrays(net_util.load_weights_npz())
    st = FusionStream(model, scene, cfg, syn.Intrinsic(), dev, a.frames, deg_per_frame=0.5)
    for i in range(a.frames):
        st.step(i, "none")
    out = {"map_voxels": int(st.map.n_occupied)}
    for name, scale in (("tracker cloud 320x240", 0.5), ("full frame 640x480", 1.0)):
        intr = syn.Intrinsic().scaled(scale)
        xyz, _ = syn.frame_points(scene, a.frames // 2, intr, device=dev)
        q = xyz.contiguous()
        with torch.no_grad():
            sdf, std, mask = st.map.get_sdf(q)
        m = int(mask.sum())
        t_val = timed(lambda: st.map.get_sdf(q))

        def with_grad():
            x = q.clone().requires_grad_(True)
            s, sd, mk = st.map.get_sdf(x)
            (g,) = torch.autograd.grad((s / sd.detach()).sum(), x)
            return g
        t_grad = timed(with_grad)

        def direct():                      # the same numbers without torch's autograd engine: the kernel's own Jacobian
            s, sd, mk, g = st.map.get_sdf_with_gradient(q)
            return g / sd.unsqueeze(1)
        t_direct = timed(direct)
        g_auto, g_direct = with_grad(), direct()
        assert torch.allclose(g_auto[mask], g_direct, rtol=1e-6, atol=1e-6)
        out[name] = {"points": int(q.shape[0]), "valid_points": m, "values_ms": round(t_val, 4), "values_and_gradient_ms": round(t_grad, 4),
                     "values_and_gradient_without_autograd_engine_ms": round(t_direct, 4),
                     "values_algorithmic_tflops": round(m * DEC_FLOP_PER_ROW / (t_val * 1e-3) / 1e12, 2)}
    # one Gauss-Newton iteration of the SDF term, the tracker's 320 x 240 cloud of the middle frame against the map, from the previous frame's pose
    from di_fusion_amd.system.tracker import Pose, sdf_hg
    f = a.frames // 2
    obs, R, t = syn.frame_cloud_camera(scene, f, syn.Intrinsic().scaled(0.5), device=dev)
    last, delta = Pose(*syn.orbit_pose(f - 1)), Pose()
    Hr, gr, er = reference_iteration(st.map, obs, last, delta)
    H, g, e, M = sdf_hg(st.map, obs, last, delta, "huber", 5.0)
    assert np.abs(H - Hr).max() < 1e-4 * np.abs(Hr).max() and abs(e - er) < 1e-5 * max(1.0, er)
    out["gauss_newton_iteration 320x240"] = {
        "points": int(obs.size(0)), "valid_points": M,
        "reference_op_sequence_ms": round(wall(lambda: reference_iteration(st.map, obs, last, delta)), 4),
        "one_call_ms": round(wall(lambda: sdf_hg(st.map, obs, last, delta, "huber", 5.0)), 4),
        "one_call_no_grad_ms": round(wall(lambda: sdf_hg(st.map, obs, last, delta, "huber", 5.0, no_grad=True)), 4),
        "clock": "host wall clock per iteration, H / g / energy on the host at the end of each"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
