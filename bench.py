#!/usr/bin/env python3
"""bench.py — frames/s of integrate + decode + mesh on a synthetic 640x480 depth stream (BASELINE.json metric).

One process per GPU (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`); every rank fuses its own
subsequence of the orbit into its own map (weak scaling, no data-path collective; SURVEY.md section 8e "C4").
A "step" = one frame: unproject+transform -> integrate_keyframe -> extract_mesh (decode, marching cubes, D2H, mesh cache).
Inputs (depth + camera-frame normals) are rendered before the timed region and stay resident in HBM.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

ENC_FLOP_PER_ROW = 52096          # encoder FLOP per gathered point
DEC_FLOP_PER_ROW = 98816          # decoder FLOP per sample row
PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md:41


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c3", choices=["c1", "c2", "c3"])
    ap.add_argument("--noise", type=int, default=0)
    ap.add_argument("--d2h", default="new", choices=["none", "new", "full"], help="what leaves the GPU each frame")
    ap.add_argument("--pipeline", type=int, default=1, help="1: enqueue frame i before completing frame i-1 on the host (no GPU idle at frame boundaries)")
    ap.add_argument("--graph", type=int, default=1, help="1: replay the frame's launches from a captured hipGraph; every --sample-every-th frame runs "
                    "eagerly with HIP events around the MFMA kernels (the roofline sample)")
    ap.add_argument("--sample-every", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the extra run with the mesh left in HBM (keeps profiler traces to one stream)")
    ap.add_argument("--cpu-sample-scale", type=float, default=0.5, help="image scale of the CPU-baseline sample frame")
    return ap.parse_args()


def cpu_baseline(cfg_name, scale):
    """The oracle (numpy port of the reference path + C marching cubes) on ONE subsampled frame of the same workload."""
    from di_fusion_amd import synthetic as syn
    from di_fusion_amd.network import utility as net_util
    from oracle import difusion_oracle as O
    scene, cfg = getattr(syn, f"config_{cfg_name}")()
    intr = syn.Intrinsic().scaled(scale)
    xyz, nrm = syn.frame_points(scene, 0, intr)
    om = O.OracleMap(O.OracleNetworks(net_util.load_weights_npz()), cfg.bound_min, cfg.bound_max, cfg.voxel_size)
    O.build_mc_oracle()
    t0 = time.perf_counter()
    om.integrate_keyframe(xyz.numpy(), nrm.numpy())
    t1 = time.perf_counter()
    om.extract_mesh(4, int(4e6), max_std=0.15)
    t2 = time.perf_counter()
    frac = (intr.width * intr.height) / (640 * 480)
    return {"value": round(frac / (t2 - t0), 5), "unit": "frames/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"frame 0 of the {cfg_name} stream rendered at {intr.width}x{intr.height} ({frac:.3f} of the pixels, "
                      f"{xyz.shape[0]} points): oracle integrate {t1 - t0:.2f}s + decode/MC {t2 - t1:.2f}s; value = pixel fraction / time; "
                      "numpy/OpenBLAS matmuls use all host cores, the rest is single-threaded"}


def flush_c_stdio():
    """RCCL prints a version banner through C stdio, which is block-buffered when stdout is a pipe and would otherwise surface after
    (or in the middle of) the JSON line."""
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


WORKLOADS = {"c1": "C1 32^3 grid 0.1 m, sphere", "c2": "C2 64^3 grid 0.1 m, room",
             "c3": "C3 128^3 grid 0.05 m, ScanNet-shape 6 m room with boxes"}


def prime_process(FusionStream, syn, model, intr, dev, d2h):
    """One-time process costs (code-object load, kernel attributes, pinned-memory pools, graph machinery) are paid on a throwaway 32^3
    map, so that they do not land in the timed region when the caller asks for little or no warmup.  Its allocator blocks are given
    back: leaving them cached costs 9 % of throughput (buffer placement matters at 0.35 ms per frame)."""
    import gc
    s1, c1 = syn.config_c1()
    prime = FusionStream(model, s1, c1, intr, dev, 4, deg_per_frame=0.5)
    prime.step(0, d2h)
    prime.step_pipelined(1, d2h)
    prime.step_graph(2, d2h)
    prime.step_graph(3, d2h)
    prime.flush(d2h)
    torch.cuda.synchronize()
    del prime
    gc.collect()
    torch.cuda.empty_cache()


def frame_runner(stream, a, d2h):
    """(run(i), drain()) for the chosen way of driving a frame."""
    def run(i):
        if a.graph and i >= 2 and (i % a.sample_every) != 0:
            return stream.step_graph(i, d2h)
        return stream.step_pipelined(i, d2h) if (a.pipeline or a.graph) else stream.step(i, d2h)

    def drain():
        stream.flush(d2h)
    return run, drain


def rate_with_mesh_left_in_hbm(make_stream, a, n_frames):
    """Secondary figure (N=1 only, reported next to `value`, never instead of it): the same stream without the per-frame hand-over of
    the new triangles to pinned host memory.  The difference is PCIe traffic, not kernels."""
    s2 = make_stream()
    run2, drain2 = frame_runner(s2, argparse.Namespace(**{**vars(a), "sample_every": 1 << 30}), "none")
    for i in range(a.warmup):
        run2(i)
    drain2()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for i in range(a.warmup, n_frames):
        run2(i)
    drain2()
    torch.cuda.synchronize()
    return round(a.steps / (time.perf_counter() - t2), 3)


def global_map_merge(stream, model, cfg, dev, barrier):
    """BASELINE config C4: after the independent subsequences, ONE all-gather of voxel records over RCCL and a fold into a global map
    (identical on every rank), meshed once.  Outside the clock (once per sequence, not per frame); reported, never fatal."""
    try:
        from di_fusion_amd import parallel
        from di_fusion_amd.system.map import DenseIndexedMap
        barrier()
        tm = time.perf_counter()
        gmap = parallel.build_global_map(stream.map, lambda: DenseIndexedMap(model, cfg.namespace(), 29, dev, initial_capacity=1 << 17))
        torch.cuda.synchronize()
        t_merge = time.perf_counter() - tm
        gmesh = gmap.extract_mesh_arrays(4, int(8e6), max_std=0.15, no_cache=True, to_host=False)
        torch.cuda.synchronize()
        return {"all_gather_and_fold_ms": round(t_merge * 1e3, 2), "global_voxels": int(gmap.n_occupied),
                "local_voxels_rank0": int(stream.map.n_occupied), "global_mesh_triangles": int(gmesh[0].shape[0]) if gmesh else 0,
                "extract_global_ms": round((time.perf_counter() - tm - t_merge) * 1e3, 2)}
    except Exception as e:      # the headline number must survive a failure of the optional epilogue
        return {"error": repr(e)[:200]}


def roofline_block(prof, sst):
    """`roofline` for the MFMA kernel with the longest average launch.  prof: name -> (ms, launches) from the library's HIP events;
    sst: the counters of the frames those events bracketed (rows per launch come from there)."""
    rows = {"encode": (sum(s["M"] for s in sst), ENC_FLOP_PER_ROW), "decode_lattice": (sum(s["B"] * 64 for s in sst), DEC_FLOP_PER_ROW),
            "decode_points": (sum(s["VH"] for s in sst), DEC_FLOP_PER_ROW)}
    kern = {}
    for name, (n_rows, flop) in rows.items():
        t_ms, n = prof[name]
        if n > 0 and t_ms > 0:
            kern[name] = dict(ms_per_launch=t_ms / n, rows_per_launch=n_rows / n, tflops=n_rows * flop / (t_ms * 1e-3) / 1e12)
    if not kern:
        return None
    dom = max(kern, key=lambda k: kern[k]["ms_per_launch"])
    pmc = {}
    try:    # HBM bytes per launch from the committed rocprofv3 PMC passes (separate runs; see profiles/README.md)
        pmc = json.loads(sorted((ROOT / "profiles").glob("r*_pmc_hbm.json"))[-1].read_text())["kernels"]
    except Exception:
        pass
    kname = {"encode": "k_encode", "decode_lattice": "k_decode_voxels", "decode_points": "k_decode<false>"}[dom]
    return {"bound": "mfma", "kernel": kname,
            "achieved": round(kern[dom]["tflops"], 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(kern[dom]["tflops"] / PEAK_FP32_MFMA_TFLOPS, 4),
            "traffic": pmc.get(kname, {}).get("hbm_bytes_per_launch"),
            "avg_launch_ms": round(kern[dom]["ms_per_launch"], 4), "rows_per_launch": round(kern[dom]["rows_per_launch"], 1),
            "per_kernel": {k: {kk: round(vv, 4) for kk, vv in v.items()} for k, v in kern.items()},
            "event_timed_frames": len(sst),
            "other_ms_per_frame": {k: round(prof[k][0] / max(1, len(sst)), 4) for k in ("mc_count", "mc_emit")}}


def main():
    import gc
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("DIF_FORCE_DIST") == "1"      # DIF_FORCE_DIST: exercise the RCCL path with one rank
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        dist.barrier()                      # first collective: RCCL builds its communicator (and prints its version banner) here,
        torch.cuda.synchronize()            # not inside the timed region
        flush_c_stdio()                     # the banner sits in C stdio's buffer: push it out before any JSON is printed
    from di_fusion_amd import _lib, synthetic as syn
    from di_fusion_amd.network import utility as net_util
    from di_fusion_amd.stream import FusionStream

    scene, cfg = getattr(syn, f"config_{a.config}")()
    intr = syn.Intrinsic()
    model = net_util.networks_from_arrays(net_util.load_weights_npz())
    n_frames = a.warmup + a.steps

    def make_stream():      # every rank walks its own arc of the orbit (independent subsequence)
        return FusionStream(model, scene, cfg, intr, dev, n_frames, deg_per_frame=0.5, phase_deg=rank * 45.0, noise=bool(a.noise))

    stream = make_stream()
    lib = _lib.load()
    if os.environ.get("DIF_BENCH_NO_PRIME") != "1":
        prime_process(FusionStream, syn, model, intr, dev, a.d2h)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    run, drain = frame_runner(stream, a, a.d2h)
    for i in range(a.warmup):
        run(i)
    drain()
    if a.graph and a.warmup >= 1 and stream._graphs is None:
        torch.cuda.synchronize()
        with torch.cuda.device(dev):
            stream._graph_export = (a.d2h == "new")
            stream._capture_graphs()            # a short warmup never reached the first replay: capture outside the clock
    stats_base = len(stream.stats)
    lib.dif_profile_read((ctypes.c_double * _lib.PROF_COUNT)(), (ctypes.c_int64 * _lib.PROF_COUNT)(), 1)
    lib.dif_profile_enable(1)
    gc.collect()
    gc.disable()            # a generation-2 collection in the middle of a 70 ms timed region shows up as a 20 % outlier
    barrier()
    t0 = time.perf_counter()
    for i in range(a.warmup, n_frames):
        run(i)
    drain()
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    lib.dif_profile_enable(0)
    ms = (ctypes.c_double * _lib.PROF_COUNT)()
    nl = (ctypes.c_int64 * _lib.PROF_COUNT)()
    _lib.check(lib.dif_profile_read(ms, nl, 1), "dif_profile_read")
    if use_dist:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    hbm_resident = None
    if world == 1 and a.d2h != "none" and not a.no_secondary:
        hbm_resident = rate_with_mesh_left_in_hbm(make_stream, a, n_frames)
    merge_info = global_map_merge(stream, model, cfg, dev, barrier) if use_dist else None

    out = None
    if rank == 0:
        st = stream.stats[stats_base:]
        # frames whose kernels were bracketed by HIP events: all of them when eager, the sampled ones under hipGraph replay
        timed_idx = [j for j in range(a.steps) if not (a.graph and (a.warmup + j) >= 2 and ((a.warmup + j) % a.sample_every) != 0)]
        prof = {n: (ms[i], nl[i]) for i, n in enumerate(_lib.PROF_NAMES)}
        launch = (f"hipGraph replay, 1 frame in {a.sample_every} eager with HIP events (roofline sample)" if a.graph else "eager")
        out = {"metric": "frames/s integrate+decode+mesh, 640x480 synthetic stream", "value": round(world * a.steps / dt, 3),
               "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": WORKLOADS[a.config] + ", 640x480 orbit stream 0.5 deg/frame, all 307200 pixels integrated and meshed "
                                                            "every frame, resolution 4, fast decode, max_std 0.15",
                          "points_per_frame": intr.width * intr.height,
                          "parallelism": f"{world} independent subsequences (one map per GPU)", "d2h_per_frame": a.d2h,
                          "host_pipeline_depth": 2 if (a.pipeline or a.graph) else 1, "launch": launch,
                          "avg_per_frame": {k: round(float(np.mean([s[k] for s in st])), 1)
                                            for k in ("M", "C", "K", "B", "VH", "T", "n_occupied", "cache_T")},
                          "frames_per_s_with_mesh_left_in_hbm": hbm_resident,
                          "graph_captures": stream.n_captures, "mesh_log_compactions": stream.map._gc_epoch,
                          "global_map_merge_after_the_clock": merge_info},
               "roofline": roofline_block(prof, [st[j] for j in timed_idx])}
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(a.config, a.cpu_sample_scale)
    if use_dist:
        flush_c_stdio()
        dist.barrier()                      # every rank has flushed whatever it had to say before rank 0 prints the one JSON line
        dist.destroy_process_group()
    flush_c_stdio()
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
