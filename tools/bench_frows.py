"""Times the remaining SURVEY.md section-8f kernels that tools/bench_query.py / bench_cloud.py do not cover: the fused image-space front end
(`ext.depth_frontend`, 8f-2) on a noisy 640x480 frame and the latent optimisation stage (`integrate_keyframe(do_optimize=True)`, 8f-4) on
the C2 room.  Usage: python tools/bench_frows.py [--reps 20]"""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from di_fusion_amd import synthetic as syn                      # noqa: E402
from di_fusion_amd.network import utility as net_util            # noqa: E402
from di_fusion_amd.system import ext                              # noqa: E402
from di_fusion_amd.system.map import DenseIndexedMap              # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    out = {}
    intr = syn.Intrinsic()
    R, t = syn.orbit_pose(2, deg_per_frame=5.0)
    depth, _ = syn.render_frame(syn.default_room(), R, t, intr, noise_seed=5)
    depth = torch.nan_to_num(depth.to(dev), nan=0.0).contiguous()
    t_fe = timed(lambda: ext.depth_frontend(depth, intr.fx, intr.fy, intr.cx, intr.cy, filter=True, want_frame=True), a.reps)
    px = intr.height * intr.width
    out["depth_frontend_640x480"] = {"ms": round(t_fe, 4), "algorithmic_bytes": px * (4 + 4 + 12 + 16 + 4 + 12),
                                     "algorithmic_gb_s": round(px * 52 / (t_fe * 1e-3) / 1e9, 1)}
    # latent optimisation: a few frames of the C2 room, 5 Adam steps per integrate (the reference's optim_n_iters)
    scene, cfg = syn.config_c2()
    model = net_util.networks_from_arrays(net_util.load_weights_npz())
    ns = cfg.namespace()
    ns.optim_n_iters, ns.code_regularization, ns.code_reg_lambda = 5, True, 1e-2
    frames = [syn.frame_points(scene, f, intr, deg_per_frame=0.5, device=dev) for f in range(4)]
    rows = voxels = 0

    def run():
        nonlocal rows, voxels
        m = DenseIndexedMap(model, ns, 29, dev, initial_capacity=1 << 15)
        for xyz, nrm in frames:
            m.integrate_keyframe(xyz, nrm, do_optimize=True)
        c = m._read_counters()
        rows, voxels = c["opt_rows"], c["opt_voxels"]
    t_opt = timed(run, max(3, a.reps // 4))

    def run_plain():
        m = DenseIndexedMap(model, ns, 29, dev, initial_capacity=1 << 15)
        for xyz, nrm in frames:
            m.integrate_keyframe(xyz, nrm)
        m._read_counters()
    t_plain = timed(run_plain, max(3, a.reps // 4))
    out["optimize_c2_4_frames"] = {"ms_with_optimisation": round(t_opt, 3), "ms_without": round(t_plain, 3), "last_frame_rows": int(rows),
                                   "last_frame_voxels": int(voxels)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
