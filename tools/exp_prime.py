"""What in bench.py's prime_process slows the SDMA export afterwards?  One priming variant per process: python tools/exp_prime.py <variant>"""
import gc, sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from di_fusion_amd import _lib, synthetic as S
from di_fusion_amd.network import utility as net_util
from di_fusion_amd.stream import FusionStream

DEV = torch.device("cuda:0")
variant = sys.argv[1]
model = net_util.networks_from_arrays(net_util.load_weights_npz())
scene, cfg = S.config_c3()
F = 120
st = FusionStream(model, scene, cfg, S.Intrinsic(), DEV, F, deg_per_frame=0.5)
assert st.enable_overlap()
if variant != "none":
    s1, c1 = S.config_c1()
    prime = FusionStream(model, s1, c1, S.Intrinsic(), DEV, 4, deg_per_frame=0.5)
    d2h = "new" if "new" in variant else "dma"
    prime.step(0, d2h)
    if "pipe" in variant or variant == "all":
        prime.step_pipelined(1, d2h)
    if "graph" in variant or variant == "all":
        prime.step_graph(2, d2h)
    if "direct" in variant or variant == "all":
        prime.step_direct(3, d2h)
    prime.flush(d2h)
    torch.cuda.synchronize()
    if "keep" not in variant:
        del prime
        gc.collect()
        if "noempty" not in variant:
            torch.cuda.empty_cache()
lib = _lib.load()
times = []
orig = lib.dif_mesh_cache_export_sdma
def timed(*a):
    t0 = time.perf_counter(); rc = orig(*a); times.append((time.perf_counter() - t0) * 1e6); return rc
class Proxy:
    def __getattr__(self, k):
        return timed if k == "dif_mesh_cache_export_sdma" else getattr(lib, k)
_lib._lib = Proxy()
for i in range(20):
    (st.step_pipelined if i < 2 else st.step_direct)(i, "dma")
st.flush("dma"); torch.cuda.synchronize(); times.clear()
t0 = time.perf_counter()
for i in range(20, F):
    st.step_direct(i, "dma")
st.flush("dma"); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / (F - 20)
print(f"prime={variant:22s}: {dt * 1e6:7.1f} us/frame; sdma call median {np.median(times):6.1f} us p90 {np.percentile(times, 90):6.1f}", flush=True)
