#!/usr/bin/env python3
"""Encoder at scale: integrate a dense first frame (all 307,200 points gather into ~8 rows each = ~2.4 M encoder rows) into a fresh
map, several times.  Reports the MFMA throughput of `k_encode` with every SIMD loaded (the C3 stream's steady state only feeds
it ~70 k rows per frame).  Usage: python tools/stress_integrate.py [--reps 5] [--config c3]"""
import argparse
import ctypes
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--config", default="c1")
    a = ap.parse_args()
    from di_fusion_amd import _lib, synthetic as syn
    from di_fusion_amd.network import utility as net_util
    from di_fusion_amd.system.map import DenseIndexedMap
    dev = torch.device("cuda:0")
    model = net_util.networks_from_arrays(net_util.load_weights_npz())
    scene, cfg = getattr(syn, f"config_{a.config}")()
    xyz, nrm = syn.frame_points(scene, 0, syn.Intrinsic(), device=dev)
    lib = _lib.load()
    runs = []
    for rep in range(a.reps + 1):
        m = DenseIndexedMap(model, cfg.namespace(), 29, dev)
        m.integrate_keyframe(xyz[:1000], nrm[:1000])          # sizes the buffers / warms the kernels up; prunes to nothing
        lib.dif_profile_read((ctypes.c_double * _lib.PROF_COUNT)(), (ctypes.c_int64 * _lib.PROF_COUNT)(), 1)
        lib.dif_profile_enable(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.integrate_keyframe(xyz, nrm)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        lib.dif_profile_enable(0)
        ms = (ctypes.c_double * _lib.PROF_COUNT)(); nl = (ctypes.c_int64 * _lib.PROF_COUNT)()
        lib.dif_profile_read(ms, nl, 1)
        c = m._read_counters()
        if rep == 0:
            continue
        runs.append(dict(integrate_ms=round(dt * 1e3, 3), M=c["M"], C=c["C"], items=c["items"], n_occupied=c["n_occupied"],
                         encode_ms=round(ms[0], 4), encode_tflops=round(c["M"] * 52096 / (ms[0] * 1e-3) / 1e12, 2)))
    print(json.dumps({"workload": f"first frame of the {a.config} stream, {xyz.size(0)} points into an empty map", "runs": runs}))


if __name__ == "__main__":
    main()
