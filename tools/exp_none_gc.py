import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from di_fusion_amd import _lib, synthetic as S
from di_fusion_amd.network import utility as net_util
from di_fusion_amd.stream import FusionStream
DEV = torch.device("cuda:0")
model = net_util.networks_from_arrays(net_util.load_weights_npz())
F = 6
def make():
    cfg = S.MapConfig((-3.2, -3.2, -3.2), (3.2, 3.2, 3.2), 0.2)
    return FusionStream(model, S.default_room(), cfg, S.Intrinsic().scaled(0.25), DEV, F, deg_per_frame=6.0, initial_capacity=1 << 13)
st = make(); ref = []
for i in range(F):
    o = st.step(i, d2h="new"); torch.cuda.synchronize(); ref.append(tuple(x.clone() for x in o))
st = make()
st.step(0, "new"); torch.cuda.synchronize()
for i in range(1, F):
    if i == 3:
        st.map._gc_wanted = True
    o = st.step_direct(i, "none")
    if o is not None:
        lo = o[0].storage_offset() // 9
        print(f"step {i}: returned frame {i - 1}: rows [{lo}, {lo + o[0].shape[0]}) gc_len {st.map._gc_log_len} epoch {st.map._gc_epoch} last_counters kept/T {st.map.last_counters['cache_kept']}/{st.map.last_counters['cache_T']} T {st.map.last_counters['T']}")
        torch.cuda.synchronize()
        c = st.map._counters.cpu().tolist()
        print(f"    device after sync: CACHE_KEPT {c[_lib.C_CACHE_KEPT]} CACHE_T {c[_lib.C_CACHE_T]} T {c[_lib.C_T]}; equal to eager: {all(torch.equal(x.cpu(), y.cpu()) for x, y in zip(ref[i - 1], o))}")
        if i == 3:
            # where ARE frame 2's triangles?
            tri = st.map._cache[0]
            want = ref[2][0].to(DEV)
            n = want.shape[0]
            for start in range(0, int(c[_lib.C_CACHE_T]) - n + 1):
                if torch.equal(tri[start:start + 1], want[0:1]) and torch.equal(tri[start:start + n], want):
                    print(f"    frame 2's triangles sit at rows [{start}, {start + n})")
                    break
            else:
                print("    frame 2's triangles are nowhere contiguous in the log")
