"""Launch time of the extract kernels against the number of decoded voxels B (HIP events on the launch stream, median of reps): separates
the fixed cost of a launch (dispatch, weight staging, fold constants, drain) from the per-tile matrix time.
    python tools/sweep_decode.py [--sizes 1,64,256,512,768,1024,1536,2048,4096] [--reps 7]"""
import argparse
import ctypes
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1,64,256,512,768,1024,1536,2048,4096")
    ap.add_argument("--reps", type=int, default=7)
    a = ap.parse_args()
    from di_fusion_amd import _lib, synthetic as syn
    from di_fusion_amd.network import utility as net_util
    from di_fusion_amd.system.map import DenseIndexedMap
    dev = torch.device("cuda:0")
    lib = _lib.load()
    model = net_util.networks_from_arrays(net_util.load_weights_npz())
    _, cfg = syn.config_c3()
    g = torch.Generator().manual_seed(3)
    out = []
    for n in [int(s) for s in a.sizes.split(",")]:
        m = DenseIndexedMap(model, cfg.namespace(), 29, dev, initial_capacity=1 << 14)
        # a slab of a wall: n voxels of a plane x = 64 (neighbours in y/z are allocated too, so the batch is the dirty set itself)
        side = int(np.ceil(np.sqrt(n)))
        yy, zz = torch.meshgrid(torch.arange(side), torch.arange(side), indexing="ij")
        lin = (zz.flatten() + 10 + 128 * (yy.flatten() + 10) + 128 * 128 * 64)[:n]
        m.allocate_block(lin.to(dev))
        m._latent[:n] = torch.randn((n, 29), generator=g).to(dev) * 0.1
        m._obs[:n] = 100.0
        rows = {}
        for rep in range(a.reps + 2):
            m._dirty[:n] = 1
            m._recount_dirty()
            torch.cuda.synchronize()
            lib.dif_profile_enable(1)
            m.extract_mesh_arrays(4, int(4e6), max_std=0.15, to_host=False)
            torch.cuda.synchronize()
            lib.dif_profile_enable(0)
            ms = (ctypes.c_double * _lib.PROF_COUNT)()
            cnt = (ctypes.c_int64 * _lib.PROF_COUNT)()
            lib.dif_profile_read(ms, cnt, 1)
            if rep >= 2:
                for k, name in enumerate(_lib.PROF_NAMES):
                    if cnt[k]:
                        rows.setdefault(name, []).append(ms[k] / cnt[k] * 1e3)
        c = m.last_counters
        out.append(dict(n=n, B=c["B"], K=c["K"], VH=c["VH"], T=c["T"], **{k + "_us": round(float(np.median(v)), 2) for k, v in rows.items()}))
        print(json.dumps(out[-1]), flush=True)
        del m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
