"""Times the tracker-side point-cloud pre-processing on the GPU: remove_radius_outlier(16, 0.05),
estimate_normals(16, 0.1) and point_box_filter(0.02) (tracker.py:105-116) on a synthetic room frame at half (tracker.py:88-95) and full resolution.
Usage: python tools/bench_cloud.py [--reps 20] [--cpu-sample 4000]"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from di_fusion_amd import synthetic as S          # noqa: E402
from di_fusion_amd.system import ext              # noqa: E402


def cloud(scale, dev):
    intr = S.Intrinsic().scaled(scale)
    R, t = S.orbit_pose(2, deg_per_frame=5.0)
    depth, _ = S.render_frame(S.default_room(), R, t, intr, noise_seed=5)
    pc = ext.unproject_depth(depth.to(dev), intr.fx, intr.fy, intr.cx, intr.cy)
    pc = torch.cat([pc, torch.zeros_like(pc[..., :1])], -1).reshape(-1, 4)
    return pc[~torch.isnan(pc[:, 0])].contiguous()


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
    ap.add_argument("--cpu-sample", type=int, default=0, help="also time the exhaustive C oracle on the first N points")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    out = {}
    for name, scale in (("320x240", 0.5), ("640x480", 1.0)):
        pc = cloud(scale, dev)
        t_out = timed(lambda: ext.remove_radius_outlier(pc, 16, 0.05), args.reps)
        kept = pc[ext.remove_radius_outlier(pc, 16, 0.05)].contiguous()
        t_nrm = timed(lambda: ext.estimate_normals(kept, 16, 0.1, [0.0, 0.0, 0.0]), args.reps)
        t_knn = timed(lambda: ext.knn_search(kept, 16, 0.1), args.reps)
        nrm = ext.estimate_normals(kept, 16, 0.1, [0.0, 0.0, 0.0])
        ok = ~torch.isnan(nrm[:, 0])
        p3, n3 = kept[ok, :3].contiguous(), nrm[ok].contiguous()
        t_box = timed(lambda: ext.point_box_filter(p3, n3, 0.02), args.reps)       # (host wrapper included: two counter reads per call)
        out[name] = {"points": int(pc.shape[0]), "kept": int(kept.shape[0]), "remove_radius_outlier_ms": round(t_out, 4),
                     "estimate_normals_ms": round(t_nrm, 4), "knn16_ms": round(t_knn, 4),
                     "points_per_s_normals": round(kept.shape[0] / t_nrm * 1e3),
                     "point_box_filter_ms": round(t_box, 4), "boxes": int(ext.point_box_filter(p3, n3, 0.02)[0].shape[0])}
        if args.cpu_sample and name == "320x240":
            from oracle import difusion_oracle as O
            sub = kept[:args.cpu_sample].cpu().numpy()
            t0 = time.time()
            O.estimate_normals(sub, 16, 0.1, [0.0, 0.0, 0.0])
            out[name]["cpu_oracle_exhaustive_s"] = round(time.time() - t0, 3)
            out[name]["cpu_oracle_points"] = int(sub.shape[0])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
