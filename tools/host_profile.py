"""Host-side cost of the two Python-driven hot loops (cProfile, top cumulative): get_sdf with gradient (the tracker's call) and the
spatially tiled direct step with the loopback halo exchange.  python tools/host_profile.py [query|tiled]"""
import cProfile
import io
import pstats
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from di_fusion_amd import synthetic as syn                      # noqa: E402
from di_fusion_amd.network import utility as net_util            # noqa: E402
from di_fusion_amd.stream import FusionStream                    # noqa: E402


def report(pr, n, what):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
    print(f"==== {what}: {n} iterations")
    print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:6000])


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "query"
    dev = torch.device("cuda:0")
    scene, cfg = syn.config_c3()
    model = net_util.networks_from_arrays(net_util.load_weights_npz())
    if which == "query":
        st = FusionStream(model, scene, cfg, syn.Intrinsic(), dev, 30, deg_per_frame=0.5)
        for i in range(30):
            st.step(i, "none")
        xyz, _ = syn.frame_points(scene, 15, syn.Intrinsic().scaled(0.5), device=dev)
        q = xyz.contiguous()

        def with_grad():
            x = q.clone().requires_grad_(True)
            s, sd, mk = st.map.get_sdf(x)
            (g,) = torch.autograd.grad((s / sd.detach()).sum(), x)
            return g
        for _ in range(20):
            with_grad()
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(300):
            with_grad()
        torch.cuda.synchronize()
        pr.disable()
        report(pr, 300, "get_sdf with gradient, 76.8k points")
    else:
        n = 260
        st = FusionStream(model, scene, cfg, syn.Intrinsic().scaled(2.0), dev, n, deg_per_frame=0.5, tiling=(4, 8, None), halo_loopback=True,
                          initial_capacity=1 << 18)
        for i in range(2):
            st.step(i, "new")
        for i in range(2, 60):
            st.step_direct(i, "new")
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        for i in range(60, n):
            st.step_direct(i, "new")
        st.flush("new")
        torch.cuda.synchronize()
        pr.disable()
        report(pr, n - 60, "tiled step_direct with loopback halo exchange (slab 4 of 8, 1280x960)")


if __name__ == "__main__":
    main()
