import ctypes, os, sys, json
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from di_fusion_amd import synthetic as syn
from di_fusion_amd.network import utility as net_util
from di_fusion_amd.stream import FusionStream
dev = torch.device("cuda:0")
scene, cfg = syn.config_c3()
model = net_util.networks_from_arrays(net_util.load_weights_npz())
frames = int(sys.argv[1])
st = FusionStream(model, scene, cfg, syn.Intrinsic(), dev, frames, deg_per_frame=0.5)
lib = ctypes.CDLL(os.environ["DIF_LIB"])
for i in range(frames):
    st.step(i, "none")
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (2048 * 8))()
assert lib.dif_trace_read(buf, 2048 * 8) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 8).astype(np.int64)
t0 = t[:, 0].min()
us = lambda x: (x - t0) / 100.0
pc = lambda x: [round(float(np.percentile(x, q)), 1) for q in (0, 10, 50, 90, 100)]
print("B", st.stats[-1]["B"], "VH", st.stats[-1]["VH"])
print("entry", pc(us(t[:, 0])))
print("enter R", pc(us(t[:, 1])))
has = t[:, 5] > 0
print("waves with tiles", int(has.sum()), "tiles", int(t[:, 5].sum()), "max per wave", int(t[:, 5].max()), "polls", pc(t[:, 7]))
print("first tile valid", pc(us(t[has, 2])), "wait after entering R", pc((t[has, 2] - t[has, 1]) / 100.0))
print("first tile done", pc(us(t[has, 3])), "tile time", pc((t[has, 3] - t[has, 2]) / 100.0))
print("exit", pc(us(t[:, 4])), "after last own work", pc((t[has, 4] - t[has, 3]) / 100.0))
pub = t[:, 6] > t0
print("last publish", pc(us(t[pub, 6])))
for w in range(8):
    sel = np.arange(2048) // 256 == w
    print("wid", w, "enterR", pc(us(t[sel, 1])), "tiles", int(t[sel, 5].sum()), "exit", pc(us(t[sel, 4])))
