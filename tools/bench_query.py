"""Times DenseIndexedMap.get_sdf on the C3 room map: values only and values + analytic d sdf / d xyz (what one
Gauss-Newton iteration of the tracker asks for, reference tracker.py:174-218).  Usage: python tools/bench_query.py [--frames 30]"""
import argparse
import json
import sys
from pathlib import Path

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
    print(json.dumps(out))


if __name__ == "__main__":
    main()
