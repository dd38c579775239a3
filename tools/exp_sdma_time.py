"""How long does the SDMA export call take inside a running stream, and where?  Times the ctypes call per frame (C3, 60 frames)."""
import ctypes, sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from di_fusion_amd import _lib, synthetic as S
from di_fusion_amd.network import utility as net_util
from di_fusion_amd.stream import FusionStream

DEV = torch.device("cuda:0")
model = net_util.networks_from_arrays(net_util.load_weights_npz())
scene, cfg = S.config_c3()
F = 60
st = FusionStream(model, scene, cfg, S.Intrinsic(), DEV, F, deg_per_frame=0.5)
lib = _lib.load()
orig = lib.dif_mesh_cache_export_sdma
times, sizes = [], []


def timed(buf, lo, n, a, b, c):
    t0 = time.perf_counter()
    rc = orig(buf, lo, n, a, b, c)
    times.append((time.perf_counter() - t0) * 1e6); sizes.append(int(n))
    return rc


class Proxy:
    def __getattr__(self, k):
        return timed if k == "dif_mesh_cache_export_sdma" else getattr(lib, k)


_lib._lib = Proxy()
for i in range(F):
    (st.step_pipelined if i < 2 else st.step_direct)(i, "dma")
st.flush("dma")
t = np.asarray(times); s = np.asarray(sizes)
print(f"sdma export calls: {len(t)}; us per call: median {np.median(t):.1f} p10 {np.percentile(t, 10):.1f} p90 {np.percentile(t, 90):.1f}; triangles median {np.median(s):.0f}")
print("last 10:", [(int(a), round(float(b), 1)) for a, b in zip(s[-10:], t[-10:])])
# idle GPU: the same call again on the last frame's rows
torch.cuda.synchronize()
b = st.map._cache_struct()
sl = st._d_slots[0]
for n in (1000, 8000, 30000):
    tt = []
    for _ in range(20):
        t0 = time.perf_counter(); orig(ctypes.byref(b), 0, n, sl["out_ptr"][0], sl["out_ptr"][1], sl["out_ptr"][2]); tt.append((time.perf_counter() - t0) * 1e6)
    print(f"idle GPU, {n} triangles: median {np.median(tt):.1f} us")
