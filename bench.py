#!/usr/bin/env python3
"""bench.py — frames/s of integrate + decode + mesh on a synthetic 640x480 depth stream (BASELINE.json metric).

One process per GPU.  Under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` the ranks come from the
environment; a bare `python bench.py --gpus N` (N > 1) starts that launcher itself; a WORLD_SIZE that disagrees with --gpus is an error.
  --mode c4 (default): every rank fuses its own subsequence of the orbit into its own map (weak scaling, no data-path collective;
                       SURVEY.md section 8e "C4"; one all-gather merge of the maps after the clock);
  --mode tiled       : BASELINE config C5 — ONE 1280x960 stream, the grid cut into N x-slabs, halo exchange with the ring neighbours
                       (RCCL send/recv) after every integrate (strong scaling: the work per frame is fixed).
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
PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md:41   (v_mfma_f32_32x32x2_f32)
PEAK_BF16_MFMA_TFLOPS = 2500.0    # /opt/skills/guides/MI355X_MICROARCH.md:42   (v_mfma_f32_32x32x16_bf16, dense)
# bf16-pipe kernels (mlp.hip.h "x6"): an fp32 product = six exact bf16 slice products, so the matrix-pipe ceiling for the reference's fp32
# arithmetic is 2,500 / 6 TFLOP/s.  Matrix-pipe cycles one 32-row tile really occupies (per SIMD, 2.4 GHz), for the pipe-busy fraction:
#   folded decoder tile: 480 bf16 MFMAs x 32 cycles + 16 f32 MFMAs x 64;   encoder tile: 312 x 32 + 3 x 64
PEAK_X6_FP32_EQUIV_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0
TILE_PIPE_CYCLES = {"bf16x6": {"encode": 312 * 32 + 3 * 64, "decode_lattice": 480 * 32 + 16 * 64, "decode_points": 480 * 32 + 16 * 64},
                    "f32": {"encode": 419 * 64, "decode_lattice": 656 * 64, "decode_points": 656 * 64}}
SIMDS, SHADER_HZ = 1024, 2.4e9


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c3", choices=["c1", "c2", "c3"])
    ap.add_argument("--mode", default="c4", choices=["c4", "tiled"], help="c4: independent subsequences per GPU; tiled: one 1280x960 stream "
                    "spatially tiled across the GPUs (BASELINE config C5)")
    ap.add_argument("--loopback", type=int, default=0, help="--mode tiled on ONE GPU: this process plays the middle slab of S and exchanges halos with "
                    "itself (device copies of the size that would go over xGMI): export / merge kernels and message sizes land in the timed region")
    ap.add_argument("--halo", default="delta", choices=["delta", "full"], help="--mode tiled: bounded delta halo messages (default) or whole boundary layers")
    ap.add_argument("--noise", type=int, default=0)
    ap.add_argument("--d2h", default="auto", choices=["auto", "none", "new", "dma", "full"], help="what leaves the GPU each frame.  new / dma: the frame's new triangles, to pinned host memory, written by kernels (the next frame's first ones carry them) / by the copy engine beside the next frame's kernels; auto (default): dma for one directly launched stream per GPU, new for stream groups and the tiled mode")
    ap.add_argument("--pipeline", type=int, default=1, help="1 (default): direct launches (two C calls per frame), the host one frame ahead of the GPU, HIP events on "
                    "every --sample-every-th frame; 0: the façade's own synchronous calls, frame by frame")
    ap.add_argument("--sample-every", type=int, default=8)
    ap.add_argument("--mlp-pipe", choices=["bf16x6", "f32"], default=None,
                    help="matrix pipe of the MLP tiles: bf16x6 (default; fp32 products as six exact bf16 slice products) or f32 (f32-input MFMA); "
                         "same as the DIF_DECODER_PIPE environment variable")
    ap.add_argument("--streams-per-gpu", type=int, default=0, help="S >= 1: every rank fuses S independent subsequences (S private maps) whose frames share "
                    "their twelve launches (dif_integrate_frames + dif_extract_streams): `value` is then the aggregate over all world x S streams.  "
                    "0 (default): ONE stream per GPU is the measured configuration, and at N = 1 the aggregate rates for S = 2, 4, 8 are reported beside "
                    "it (config.frames_per_s_with_S_streams_per_gpu, roofline.by_streams)")
    ap.add_argument("--overlap", type=int, default=1, choices=[0, 1], help="1 (default): ONE directly launched stream uses TWO hardware queues — frame i+1's integrate "
                    "front end (unproject ... encoder) runs beside frame i's extract, ordered by device-side waits (FusionStream.enable_overlap; d2h dma / "
                    "none); 0: every frame's twelve launches on one queue.  config.two_queues says what ran")
    ap.add_argument("--group-d2h", default="new", choices=["new", "dma"], help="how the secondary S-streams-per-GPU legs of a `--d2h dma` run deliver their triangles: "
                    "new (default: carried by the next group frame's point kernels) or dma (one SDMA call per stream and group frame: equal at S = 4, 6 % behind at "
                    "S = 8 — the host)")
    ap.add_argument("--rccl-before-clock", type=int, default=0, choices=[0, 1], help="--mode c4 under torch.distributed: 0 (default) = the barriers around "
                    "the clock go over gloo and RCCL is brought up BEHIND the clock, for the exchange step (the global map merge); 1 = RCCL is the process group "
                    "from the start, alive during the timed region like in a deployment that merges maps periodically.  The line says which "
                    "(config.rccl_before_clock); tools/gpu_scale.sh runs both.  --mode tiled exchanges halos inside every frame: always 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the extra run with the mesh left in HBM (keeps profiler traces to one stream)")
    ap.add_argument("--cpu-frames", type=int, default=5, help="frames timed per thread setting by the CPU baseline (after 2 warm-ups)")
    return ap.parse_args()


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(stream, cfg_name, scene, cfg, intr, first_frame, n_timed, deg_per_frame=0.5):
    """the parity-checked CPU restatement (the numpy oracle + the C marching cubes; the reference's Python cannot
    travel to this box) on FULL-resolution frames of the same stream, continuing from the map state the GPU run reached after its
    warm-up frames (copied into the oracle before the clock, so the CPU frames are the steady-state frames the GPU number is quoted on,
    not the heavy map-building ones).  Per thread setting (1 thread, 16 threads, all host cores): 2 warm-up frames, then the median of
    `n_timed` frames, stages timed separately; `value` is the best setting's end-to-end rate and `cores` the threads it used (on a
    256-core host the small per-frame matmuls run fastest on ONE thread).  Baseline only — says nothing about kernel quality."""
    import threadpoolctl
    from di_fusion_amd import synthetic as syn
    from di_fusion_amd.network import utility as net_util
    from oracle import difusion_oracle as O
    O.build_mc_oracle()
    net = O.OracleNetworks(net_util.load_weights_npz())
    m = stream.map
    n = m.n_occupied
    state = dict(indexer=m.indexer.cpu().numpy().reshape(-1).copy(), pos=m.latent_vecs_pos[:n].cpu().numpy().copy(),
                 w=m.voxel_obs_count[:n].cpu().numpy().copy(), z=m.latent_vecs[:n].cpu().numpy().copy())

    def fresh():
        om = O.OracleMap(net, cfg.bound_min, cfg.bound_max, cfg.voxel_size)
        cap = 1
        while cap < max(n, 1):
            cap *= 2
        om.indexer = state["indexer"].copy()
        om.latent_vecs = np.zeros((cap, 29), np.float32); om.latent_vecs[:n] = state["z"]
        om.latent_vecs_pos = -np.ones((cap,), np.int64); om.latent_vecs_pos[:n] = state["pos"]
        om.voxel_obs_count = np.zeros((cap,), np.float32); om.voxel_obs_count[:n] = state["w"]
        om.n_occupied = n
        return om

    frames = []
    for i in range(first_frame, first_frame + 2 + n_timed):
        R, t = syn.orbit_pose(i, 0.3, deg_per_frame, 0.0)
        depth, ncam = syn.render_frame(scene, R, t, intr, torch.device("cpu"))
        frames.append((depth.numpy(), ncam.numpy().reshape(-1, 3), np.asarray(R, np.float64).astype(np.float32), np.asarray(t, np.float64).astype(np.float32)))
    cores = os.cpu_count() or 1
    legs = {}
    for threads in sorted({1, min(16, cores), cores}):
        om = fresh()
        rows = []
        with threadpoolctl.threadpool_limits(limits=threads):
            for k, (depth, ncam, R, t) in enumerate(frames):
                t0 = time.perf_counter()
                pc = O.unproject_depth(depth, intr.fx, intr.fy, intr.cx, intr.cy).reshape(-1, 3)          # a1
                ok = ~np.isnan(pc[:, 0])
                xyz = (pc[ok] @ R.T + t).astype(np.float32)                                               # a2
                nrm = (ncam[ok] @ R.T).astype(np.float32)
                t1 = time.perf_counter()
                om.integrate_keyframe(xyz, nrm)                                                           # a3-a10
                t2 = time.perf_counter()
                a = om.extract_prepare(4)                                                                 # a11-a14
                t3 = time.perf_counter()
                if a is not None:                                                                         # a15
                    O.marching_cubes_interp(a["indexer"], a["valid_blocks"], a["vec_batch_mapping"], a["cube_sdf"], a["cube_std"], int(4e6), om.n_xyz, 0.15)
                    om.updated_vec_id = np.zeros((0,), np.int64)
                t4 = time.perf_counter()
                if k >= 2:
                    rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0))
        med = np.median(np.asarray(rows), axis=0)
        legs[threads] = dict(zip(("unproject_s", "integrate_s", "decode_s", "marching_cubes_s", "frame_s"), (round(float(v), 4) for v in med)))
    best = min(legs, key=lambda k: legs[k]["frame_s"])
    return {"value": round(1.0 / legs[best]["frame_s"], 4), "unit": "frames/s", "cores": best, "kind": "port",
            "cpu_model": cpu_model_name(), "host_cores": cores,
            "sample": f"frames {first_frame + 2}..{first_frame + 1 + n_timed} of the {cfg_name} stream at {intr.width}x{intr.height} (all pixels), oracle map "
                      f"initialised from the GPU map after {first_frame} frames ({n} voxels); median of {n_timed} frames after 2 warm-ups per thread "
                      "setting; numpy/OpenBLAS threads limited with threadpoolctl, the C marching cubes and the index arithmetic are single-threaded",
            "per_thread_setting": {str(k): v for k, v in legs.items()}}


def flush_c_stdio():
    """RCCL prints a version banner through C stdio, which is block-buffered when stdout is a pipe and would otherwise surface after
    (or in the middle of) the JSON line."""
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


WORKLOADS = {"c1": "C1 32^3 grid 0.1 m, sphere", "c2": "C2 64^3 grid 0.1 m, room",
             "c3": "C3 128^3 grid 0.05 m, ScanNet-shape 6 m room with boxes"}


def prime_process(FusionStream, syn, model, intr, dev, d2h, overlap=False):
    """One-time process costs (code-object load, kernel attributes, pinned-memory pools) are paid on a throwaway 32^3
    map, so that they do not land in the timed region when the caller asks for little or no warmup.  The throwaway map's blocks stay in torch's caching allocator:
    until round 5 they were returned to the driver (`torch.cuda.empty_cache()`: buffer placement was worth 9 % at 0.35 ms per frame), but a
    process that has handed gigabytes back to the driver runs every later SDMA copy 6 times slower (hsa_amd_memory_async_copy of a frame's
    rows 34 -> 223 us, measured with tools/exp_prime.py: any three primed frames followed by empty_cache; without the empty_cache, or
    with the blocks kept, 34 us) — and with the blocks kept the rate is the same as without priming."""
    import gc
    s1, c1 = syn.config_c1()
    prime = FusionStream(model, s1, c1, intr, dev, 4, deg_per_frame=0.5)
    if overlap:
        prime.enable_overlap()
    prime.step(0, d2h)
    prime.step_pipelined(1, d2h)
    prime.step_direct(2, d2h)
    prime.step_direct(3, d2h)
    prime.flush(d2h)
    torch.cuda.synchronize()
    del prime
    gc.collect()
    if os.environ.get("DIF_BENCH_PRIME_EMPTY_CACHE") == "1":        # (experiments: round 4's behaviour)
        torch.cuda.empty_cache()


def frame_runner(stream, a, d2h):
    """(run(i), drain()) for the chosen way of driving a frame."""
    lib = None

    def run(i):
        nonlocal lib
        if i < 2 or not a.direct:
            return stream.step_pipelined(i, d2h) if a.pipeline else stream.step(i, d2h)
        sampled = (i % a.sample_every) == 0
        # direct launches (two C calls); on the sampled frames of the timed region with HIP events around the MFMA / marching-cubes kernels
        timed = sampled and a.timed_from is not None and i >= a.timed_from
        if timed:
            if lib is None:
                from di_fusion_amd import _lib
                lib = _lib.load()
            lib.dif_profile_enable(1)
        out = stream.step_direct(i, d2h)
        if timed:
            lib.dif_profile_enable(0)
        return out

    def drain():
        stream.flush(d2h)
    return run, drain


class GroupBench:
    """S streams of one rank driven as a `FusionStreamGroup`: frame 0 eagerly per stream (sizes the buffers), every later frame with two C
    calls for the whole group; on the sampled frames of the timed region with HIP events around the MFMA / marching-cubes launches."""

    def __init__(self, streams, a, d2h, lib):
        from di_fusion_amd.stream import FusionStreamGroup
        self.streams, self.a, self.d2h, self.lib = streams, a, d2h, lib
        self.group = FusionStreamGroup(streams)

    def run(self, i):
        a = self.a
        if i == 0:
            for st in self.streams:
                st.step(0, self.d2h)
            return
        timed = (i % a.sample_every) == 0 and a.timed_from is not None and i >= a.timed_from
        if timed:
            self.lib.dif_profile_enable(1)
        self.group.step(i, self.d2h)
        if timed:
            self.lib.dif_profile_enable(0)

    def drain(self):
        for st in self.streams:
            st.flush(self.d2h)

    def frame_stats(self, first):
        """Counters per frame from frame `first` on, summed over the streams (rows per LAUNCH)."""
        n = min(len(st.stats) for st in self.streams)
        return [{k: sum(st.stats[f][k] for st in self.streams) for k in ("M", "C", "K", "B", "VH", "T", "n_occupied", "cache_T")} for f in range(first, n)]


def timed_run(run, drain, a, n_frames, lib, barrier):
    """Warm-up frames, then EXACTLY a.steps frames between two barriers; returns (seconds, [(kernel, ms)] event records in launch order)."""
    import gc
    for i in range(a.warmup):
        run(i)
    drain()
    cap = 1 << 16
    p_which, p_ms = (ctypes.c_int32 * cap)(), (ctypes.c_float * cap)()
    lib.dif_profile_dump(p_which, p_ms, cap, 1)
    lib.dif_profile_enable(1)               # (fills the library's event pool outside the clock)
    lib.dif_profile_enable(0)
    a.timed_from = a.warmup
    gc.collect()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    for i in range(a.warmup, n_frames):
        run(i)
    drain()
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    lib.dif_profile_enable(0)
    n_rec = int(lib.dif_profile_dump(p_which, p_ms, cap, 1))
    if n_rec < 0:
        raise SystemExit("dif_profile_dump failed")
    from di_fusion_amd import _lib
    recs = [((_lib.PROF_NAMES[p_which[k]] if p_which[k] < len(_lib.PROF_NAMES) else "?"), float(p_ms[k])) for k in range(n_rec)]
    return dt, recs


def split_frames(recs):
    """Event records in launch order -> one list per event-timed frame (a frame starts with its encoder launch)."""
    per_frame = []
    for name, ms in recs:
        if name == "encode" or not per_frame:
            per_frame.append([])
        per_frame[-1].append((name, ms))
    return per_frame


def streams_leg(make_stream_j, S, a, n_frames, lib, pipe):
    """Aggregate rate of S independent subsequences on this GPU sharing their launches (secondary figure at N = 1)."""
    import gc
    streams = [make_stream_j(j) for j in range(S)]
    aa = argparse.Namespace(**{**vars(a), "timed_from": None})
    gb = GroupBench(streams, aa, a.group_d2h if a.d2h == "dma" else a.d2h, lib)
    dt, recs = timed_run(gb.run, gb.drain, aa, n_frames, lib, torch.cuda.synchronize)
    st = gb.frame_stats(a.warmup)
    timed_idx = [j for j in range(a.steps) if (a.warmup + j) >= 1 and ((a.warmup + j) % a.sample_every) == 0]
    per_frame = split_frames(recs)
    blk = None
    if len(per_frame) == len(timed_idx) and per_frame:
        blk = roofline_block(per_frame, [st[j] for j in timed_idx], pipe, short_run=(a.steps <= 30), streams=S)
    rate = round(S * a.steps / dt, 3)
    del gb, streams
    gc.collect()
    torch.cuda.empty_cache()
    keep = ("kernel", "achieved", "frac", "frac_executed", "pmc_mfma_busy_frac", "pmc_mfma_busy_source", "avg_launch_ms", "rows_per_launch", "per_kernel",
            "other_ms_per_frame", "event_timed_frames")
    return rate, ({k: blk[k] for k in keep if k in blk} if blk else None)


def secondary_rate(make_stream, a, n_frames, d2h):
    """Secondary figure (N=1 only, reported next to `value`, never instead of it): the same stream without the per-frame hand-over
    of the new triangles to pinned host memory — the difference is PCIe traffic, not kernels.  Best of two
    passes over fresh streams: these 40 ms measurements are informational, and a single host or driver stall (seen in about one run in
    ten on shared boxes) would otherwise decide them."""
    return max(_secondary_pass(make_stream, a, n_frames, d2h) for _ in range(2))


def _secondary_pass(make_stream, a, n_frames, d2h):
    s2 = make_stream()
    run2, drain2 = frame_runner(s2, argparse.Namespace(**{**vars(a), "sample_every": 1 << 30, "timed_from": None}), d2h)
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


def within(seconds, fn, what, dev):
    """fn() on a worker thread, waited for at most `seconds`: the epilogue behind the clock talks over a transport that this repository has
    only ever run with one rank (RCCL) — a collective that never returns must not take the measured line with it.  Returns (done, value)."""
    import threading
    box = {}

    def work():
        try:
            if dev.type == "cuda":
                torch.cuda.set_device(dev)
            box["value"] = fn()
        except Exception as e:              # (reported by the caller)
            box["value"] = {"error": repr(e)[:200]}

    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        return False, {"error": f"{what}: no answer within {seconds} s (left behind; the process exits without the final barrier)"}
    return True, box.get("value")


def collective_probe(group, device):
    """How many ranks a process group REALLY connects: every rank contributes rank + 1 to an all-reduce of a tensor on `device`; the group's
    world size if the sum is world (world + 1) / 2 on this rank, 0 otherwise (or on any failure).  `rccl_ranks` of the JSON line is this
    number for the RCCL group — observed, not the launcher's WORLD_SIZE."""
    import torch.distributed as dist
    try:
        n = dist.get_world_size(group)
        r = dist.get_rank(group)
        t = torch.full((4,), float(r + 1), dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        if t.is_cuda:
            torch.cuda.synchronize(t.device)
        want = n * (n + 1) / 2.0
        return n if bool((t.cpu() == want).all()) else 0
    except Exception:
        return 0


def global_map_merge(local_maps, model, cfg, dev, barrier, probe_box=None):
    """BASELINE config C4: after the independent subsequences, ONE all-gather of voxel records over RCCL and a fold into a global map
    (identical on every rank), meshed once.  Outside the clock (once per sequence, not per frame); reported, never fatal."""
    try:
        import torch.distributed as dist
        from di_fusion_amd import parallel
        from di_fusion_amd.system.map import DenseIndexedMap
        grp = None
        if dist.get_backend() == "gloo" and os.environ.get("DIF_BENCH_REHEARSAL") != "1":
            grp = dist.new_group(backend="nccl")                     # RCCL comes up here, behind the clock (see main)
            dist.barrier(group=grp, device_ids=[dev.index])          # (its first collective builds the communicator)
            torch.cuda.synchronize()
            flush_c_stdio()
            if probe_box is not None:
                probe_box["rccl_ranks"] = collective_probe(grp, dev)
                probe_box["when"] = "behind the clock (dist.new_group('nccl') in the map merge)"
        barrier()
        tm = time.perf_counter()
        gmap = parallel.build_global_map(local_maps, lambda: DenseIndexedMap(model, cfg.namespace(), 29, dev, initial_capacity=1 << 17), group=grp)
        torch.cuda.synchronize()
        t_merge = time.perf_counter() - tm
        gmesh = gmap.extract_mesh_arrays(4, int(8e6), max_std=0.15, no_cache=True, to_host=False)
        torch.cuda.synchronize()
        return {"all_gather_and_fold_ms": round(t_merge * 1e3, 2), "global_voxels": int(gmap.n_occupied),
                "local_voxels_rank0": int(sum(m.n_occupied for m in (local_maps if isinstance(local_maps, (list, tuple)) else [local_maps]))), "global_mesh_triangles": int(gmesh[0].shape[0]) if gmesh else 0,
                "extract_global_ms": round((time.perf_counter() - tm - t_merge) * 1e3, 2)}
    except Exception as e:      # the headline number must survive a failure of the optional epilogue
        return {"error": repr(e)[:200]}


def halo_summary(stream, a):
    """--mode tiled: what the per-frame halo refresh moved (bytes per direction and frame, message kinds) over the last frames of the run."""
    hist = getattr(stream, "_halo_buffers", {}).get("hist") if stream.tiling is not None else None
    if not hist:
        return None
    frames = sorted(hist)
    kinds = [k for f in frames for d in hist[f]["kinds"] for k in d.values()]
    recs = [int(hist[f]["out"][4 * k]) for f in frames for k in (0, 1)]
    return {"mode": a.halo, "loopback_slabs": a.loopback or None, "frames_summarised": len(frames),
            "bytes_sent_per_frame": round(float(np.mean([hist[f]["bytes_out"] for f in frames])), 1),
            "bytes_received_per_frame": round(float(np.mean([hist[f]["bytes_in"] for f in frames])), 1),
            "delta_messages_share": round(kinds.count("delta") / max(1, len(kinds)), 3),
            "records_per_message_avg": round(float(np.mean(recs)), 1), "records_per_message_max": int(max(recs)),
            "whole_layer_message_bytes": (1 + stream.map.halo_message_rows(3)) * 128}


def roofline_of(records, sst):
    """records: [(kernel name, ms)] of the event-timed launches of the frames whose counters are `sst`."""
    rows = {"encode": (sum(s["M"] for s in sst), ENC_FLOP_PER_ROW), "decode_lattice": (sum(s["B"] * 64 for s in sst), DEC_FLOP_PER_ROW),
            "decode_points": (sum(s["VH"] for s in sst), DEC_FLOP_PER_ROW)}
    kern = {}
    for name, (n_rows, flop) in rows.items():
        ts = [ms for k, ms in records if k == name]
        if ts and sum(ts) > 0:
            kern[name] = dict(ms_per_launch=sum(ts) / len(ts), rows_per_launch=n_rows / len(ts), tflops=n_rows * flop / (sum(ts) * 1e-3) / 1e12)
    return kern


def roofline_block(per_frame, sst, pipe, short_run=False, streams=1):
    """`roofline` for the MFMA kernel with the longest average launch.  per_frame: for every event-timed frame the [(kernel, ms)] list
    from the library's HIP events (recorded on the launch stream); sst: the counters of those frames (rows per launch come from there).
    The same figures are also given for the first and the second half of the timed frames: a short run sits in the map-building
    transient (thousands of voxels decoded per frame), a long one ends in steady state (~900), and the fraction differs.
    pipe: "bf16x6" (default kernels: every fp32 product as six bf16 slice products on the bf16 matrix pipe) or "f32" (f32-input MFMA).
    `achieved` is always the reference's ALGORITHMIC fp32 FLOP (98,816 per decoder row, 52,096 per encoder row) per second; `peak` is the
    matrix-pipe ceiling for that arithmetic on the pipe in use; `pipe_busy_frac` is the share of the launch the matrix pipes really worked."""
    kern = roofline_of([r for f in per_frame for r in f], sst)
    if not kern:
        return None
    peak = PEAK_X6_FP32_EQUIV_TFLOPS if pipe == "bf16x6" else PEAK_FP32_MFMA_TFLOPS

    def busy(name, v):      # matrix-pipe cycles of the launch's tiles / (SIMDs x launch duration)
        return v["rows_per_launch"] / 32.0 * TILE_PIPE_CYCLES[pipe][name] / (SIMDS * SHADER_HZ * v["ms_per_launch"] * 1e-3)

    dom = max(kern, key=lambda k: kern[k]["ms_per_launch"])
    # HBM bytes per launch and matrix-pipe busy cycles are NOT measured by this run (PMC needs rocprofv3): they come from the committed
    # summaries of separate --pmc passes OF THE SAME INVOCATION — `--steps 20 --warmup 5` (what the driver runs: map-building transient)
    # has its own set (profiles/rNN_pmc_*_k20.json), every longer run uses the 200-step set
    pmc, pmc_file, busy_pmc, busy_file = {}, None, {}, None
    sfx = "" if streams == 1 else f"_s{streams}"
    try:
        pmc_file = sorted((ROOT / "profiles").glob(f"r*_pmc_hbm{sfx}_k20.json" if short_run else f"r*_pmc_hbm{sfx}.json"))[-1]
        pmc = json.loads(pmc_file.read_text())["kernels"]
    except Exception:
        pass
    try:
        busy_file = sorted((ROOT / "profiles").glob(f"r*_pmc_mfma{sfx}_k20.json" if short_run else (f"r*_pmc_mfma{sfx}.json" if streams > 1 else "r*_pmc_mfma_stream.json")))[-1]
        busy_pmc = json.loads(busy_file.read_text())["kernels"]
    except Exception:
        pass
    kname = {"encode": "k_encode", "decode_lattice": "k_decode_voxels", "decode_points": "k_decode_refine_x6" if pipe == "bf16x6" else "k_decode<false>"}[dom]
    if streams > 1:
        kname += "_batch"
    half = len(per_frame) // 2
    phases = {}
    if half >= 1:
        for label, lo, hi in (("first_half_of_timed_frames", 0, half), ("second_half_of_timed_frames", half, len(per_frame))):
            k = roofline_of([r for f in per_frame[lo:hi] for r in f], sst[lo:hi])
            phases[label] = {n: {"frac": round(v["tflops"] / peak, 4), "pipe_busy_frac": round(busy(n, v), 4), "avg_launch_ms": round(v["ms_per_launch"], 4),
                                 "rows_per_launch": round(v["rows_per_launch"], 1)} for n, v in k.items()}
    other = {}
    for name in ("mc_count", "mc_emit"):
        ts = [ms for f in per_frame for k, ms in f if k == name]
        other[name] = round(sum(ts) / max(1, len(per_frame)), 4)
    return {"bound": "mfma", "kernel": kname,
            "pipe": ("v_mfma_f32_32x32x16_bf16: each fp32 product as six exact bf16 slice products, fp32 accumulate" if pipe == "bf16x6"
                     else "v_mfma_f32_32x32x2_f32"),
            "achieved": round(kern[dom]["tflops"], 3), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(kern[dom]["tflops"] / peak, 4),
            "peak_source": ("2,500 TFLOP/s dense bf16 MFMA / 6 slice products per fp32 product" if pipe == "bf16x6" else "157.3 TFLOP/s f32-input MFMA"),
            "frac_of_f32_input_mfma_peak": round(kern[dom]["tflops"] / PEAK_FP32_MFMA_TFLOPS, 4),
            "frac_executed": round(busy(dom, kern[dom]), 4),
            "frac_executed_is": "matrix-pipe cycles of the launch's tiles (MFMAs x cycles each) / (1,024 SIMDs x launch time): the share of the launch the pipes worked",
            "pipe_busy_frac": round(busy(dom, kern[dom]), 4),
            "pmc_mfma_busy_frac": next((busy_pmc[k]["mfma_util"] for k in (kname, kname + ("<true>" if pipe == "bf16x6" else "<false>")) if k in busy_pmc), None),
            "pmc_mfma_busy_source": (f"static: profiles/{busy_file.name} (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE of the same invocation)"
                                     if busy_file else None),
            "traffic": next((pmc[k]["hbm_bytes_per_launch"] for k in (kname, kname + ("<true>" if pipe == "bf16x6" else "<false>")) if k in pmc), None),
            "traffic_source": (f"static: profiles/{pmc_file.name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same invocation, "
                               "FETCH_SIZE doubled as the gfx950 guide prescribes); not measured by this run") if pmc_file else None,
            "avg_launch_ms": round(kern[dom]["ms_per_launch"], 4), "rows_per_launch": round(kern[dom]["rows_per_launch"], 1),
            "per_kernel": {k: dict({kk: round(vv, 4) for kk, vv in v.items()}, frac=round(v["tflops"] / peak, 4), pipe_busy_frac=round(busy(k, v), 4))
                           for k, v in kern.items()},
            "event_timed_frames": len(sst), "by_phase": phases,
            "other_ms_per_frame": other}


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start `torch.distributed.run` with N ranks on this node and pass its output on."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    import gc
    a = parse()
    # DIF_BENCH_REHEARSAL=1: every rank on cuda:0, gloo instead of RCCL (which refuses two ranks on one device; the library stages the
    # messages through pinned host memory) — the multi-rank code paths of this file and of di_fusion_amd.parallel with live processes on
    # a ONE-GPU box (tests/test_gpu_multiproc.py).  The numbers of such a run mean nothing.
    rehearsal = os.environ.get("DIF_BENCH_REHEARSAL") == "1"
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        if torch.cuda.device_count() < a.gpus and not rehearsal:
            raise SystemExit(f"bench.py --gpus {a.gpus}: only {torch.cuda.device_count()} GPU(s) visible")
        spawn_ranks(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {a.gpus} bench.py --gpus {a.gpus} ...)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("DIF_FORCE_DIST") == "1"      # DIF_FORCE_DIST: exercise the RCCL path with one rank
    rccl_probe = {"rccl_ranks": 0, "when": None}     # filled by collective_probe() where the RCCL group comes up: observed, not asserted
    clock_over_gloo = False
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # --mode c4 has no collective inside the clock (independent subsequences): the barriers and the max-over-ranks of the time go over
        # gloo, and RCCL is brought up BEHIND the clock for the exchange step (the global map merge: dist.new_group("nccl") there).  A process
        # that has initialised RCCL runs the copy-delivered export 10 % slower (5,100 against 5,870 frames/s with one rank, measured; gloo
        # does not) — this keeps a rank of an N-GPU run on the same path, and at the same rate, as the single-GPU run.  --mode tiled
        # exchanges halos inside every frame: RCCL from the start.
        clock_over_gloo = rehearsal or (a.mode != "tiled" and not a.rccl_before_clock)
        if clock_over_gloo:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        dist.barrier()                      # first collective: RCCL builds its communicator (and prints its version banner) here,
        torch.cuda.synchronize()            # not inside the timed region
        flush_c_stdio()                     # the banner sits in C stdio's buffer: push it out before any JSON is printed
        if not clock_over_gloo:             # RCCL is the process group: observe, before the clock, how many ranks it really connects
            rccl_probe["rccl_ranks"] = collective_probe(None, dev)
            rccl_probe["when"] = "before the clock (RCCL is the process group of the run)"
    from di_fusion_amd import _lib, synthetic as syn
    from di_fusion_amd.network import utility as net_util
    from di_fusion_amd.stream import FusionStream

    tiled = a.mode == "tiled"
    if a.loopback > 1 and (not tiled or world != 1):
        raise SystemExit("bench.py --loopback S needs --mode tiled on one GPU")
    scene, cfg = getattr(syn, f"config_{a.config}")()
    intr = syn.Intrinsic().scaled(2.0) if tiled else syn.Intrinsic()          # C5: one 1280x960 stream
    model = net_util.networks_from_arrays(net_util.load_weights_npz(), x6=(None if a.mlp_pipe is None else a.mlp_pipe == "bf16x6"))
    pipe = "bf16x6" if model.packed.x6 else "f32"
    n_frames = a.warmup + a.steps
    a.direct = bool(a.pipeline)
    a.timed_from = None
    a.n_frames = n_frames

    def make_stream():
        if tiled:           # every rank renders the same stream and owns one x-slab of the grid
            tiling = (a.loopback // 2, a.loopback, None) if a.loopback > 1 else (rank, world, None)
            return FusionStream(model, scene, cfg, intr, dev, n_frames, deg_per_frame=0.5, noise=bool(a.noise),
                                tiling=tiling, halo_mode=a.halo, halo_loopback=a.loopback > 1)
        st = FusionStream(model, scene, cfg, intr, dev, n_frames, deg_per_frame=0.5, phase_deg=rank * 45.0, noise=bool(a.noise))   # own arc of the orbit
        if a.overlap and a.direct and a.d2h in ("dma", "none"):
            st.enable_overlap()             # (stays off, and says so, when no second hardware queue is to be had)
        return st

    S_main = int(a.streams_per_gpu)
    if a.d2h == "auto":
        a.d2h = "dma" if (a.direct and S_main <= 1 and not tiled) else "new"
    if S_main < 0 or S_main > _lib.MAX_STREAMS or (S_main >= 1 and tiled):
        raise SystemExit(f"bench.py --streams-per-gpu: 0..{_lib.MAX_STREAMS}, with --mode c4 and direct launches")

    def make_stream_j(j, S):
        """Stream j of S on this rank: its own arc of the orbit, its own map."""
        return FusionStream(model, scene, cfg, intr, dev, n_frames, deg_per_frame=0.5, phase_deg=((rank * S + j) * 45.0) % 360.0, noise=bool(a.noise))

    lib = _lib.load()
    gb = None
    if S_main >= 1:
        gb = GroupBench([make_stream_j(j, S_main) for j in range(S_main)], a, a.d2h, lib)
        stream = gb.streams[0]
    else:
        stream = make_stream()
    if os.environ.get("DIF_BENCH_NO_PRIME") != "1":
        prime_process(FusionStream, syn, model, syn.Intrinsic(), dev, a.d2h, overlap=bool(stream.overlap))

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    run, drain = (gb.run, gb.drain) if gb is not None else frame_runner(stream, a, a.d2h)
    for i in range(a.warmup):
        run(i)
    drain()
    stats_base = len(stream.stats)
    cap = 1 << 16
    p_which, p_ms = (ctypes.c_int32 * cap)(), (ctypes.c_float * cap)()
    lib.dif_profile_dump(p_which, p_ms, cap, 1)
    lib.dif_profile_enable(1)               # (fills the library's event pool outside the clock)
    lib.dif_profile_enable(0)
    a.timed_from = a.warmup
    if not a.direct:
        lib.dif_profile_enable(1)           # (otherwise the runner switches it on for the sampled frames only)
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
    n_rec = int(lib.dif_profile_dump(p_which, p_ms, cap, 1))
    if n_rec < 0:
        raise SystemExit("dif_profile_dump failed")
    if use_dist:
        tt = torch.tensor([dt], device=("cpu" if clock_over_gloo else dev), dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    hbm_resident = None
    # (secondary figures need a run long enough to amortise their own one-time costs — a fresh stream's buffer growth)
    if world == 1 and a.d2h != "none" and not a.no_secondary and not tiled and a.steps >= 100 and gb is None:
        hbm_resident = secondary_rate(make_stream, a, n_frames, "none")
    # S independent subsequences per GPU sharing their launches: aggregate frames/s and the MFMA kernels' roofline at S = 2, 4, 8
    by_streams = {}
    if world == 1 and not a.no_secondary and not tiled and gb is None and a.direct and a.warmup + a.steps >= 2:
        for S in (2, 4, 8):
            try:
                by_streams[S] = streams_leg(lambda j, S=S: make_stream_j(j, S), S, a, n_frames, lib, pipe)
            except Exception as e:      # the headline number must survive a failure of a secondary leg
                by_streams[S] = (None, {"error": repr(e)[:200]})
    epilogue_ok, merge_info = True, None
    if use_dist and not tiled:
        epilogue_ok, merge_info = within(float(os.environ.get("DIF_BENCH_EPILOGUE_TIMEOUT", "120")),
                                         lambda: global_map_merge([st.map for st in gb.streams] if gb is not None else stream.map, model, cfg, dev, barrier, rccl_probe),
                                         "global map merge", dev)

    out = None
    if rank == 0:
        st = gb.frame_stats(stats_base) if gb is not None else stream.stats[stats_base:]
        # frames whose kernels were bracketed by HIP events: all of them when eager, the sampled ones of a directly launched run
        sampled_only = bool(a.direct)
        if gb is not None:
            timed_idx = [j for j in range(a.steps) if (a.warmup + j) >= 1 and ((a.warmup + j) % a.sample_every) == 0]
        else:
            timed_idx = [j for j in range(a.steps) if not (sampled_only and (a.warmup + j) >= 2 and ((a.warmup + j) % a.sample_every) != 0)]
        # the event records come in launch order, one k_encode per event-timed frame: cut the list into frames there
        per_frame = []
        for k in range(n_rec):
            name = _lib.PROF_NAMES[p_which[k]] if p_which[k] < len(_lib.PROF_NAMES) else "?"
            if name == "encode" or not per_frame:
                per_frame.append([])
            per_frame[-1].append((name, float(p_ms[k])))
        if len(per_frame) != len(timed_idx):        # (an empty frame launches no encoder) fall back to one group
            per_frame, timed_idx = [[r for f in per_frame for r in f]], timed_idx[:1] if timed_idx else []
        launch = (f"direct launches, two C calls per frame of ALL {S_main} streams of the rank (dif_integrate_frames + dif_extract_streams: blockIdx.y = stream in "
                  f"the point / scan / fusion / marching-cubes kernels, concatenated tile ranges in the persistent MLP kernels), host one frame ahead, HIP events "
                  f"on 1 frame in {a.sample_every}" if gb is not None else
                  "eager, frame by frame" if not sampled_only else
                  f"direct launches (two C calls per frame), host one frame ahead, HIP events on 1 frame in {a.sample_every} (roofline sample)")
        pixels = intr.width * intr.height
        value = (a.steps if tiled else world * max(S_main, 1) * a.steps) / dt       # tiled: ONE stream, however many GPUs work on it
        out = {"metric": f"frames/s integrate+decode+mesh, {intr.width}x{intr.height} synthetic stream", "value": round(value, 3),
               "unit": "frames/s", "n_gpus": world,
               # observed: the world size of the RCCL group after an all-reduce of a device tensor over it returned the right sum on this rank; 0 if RCCL
               # never came up (or the probe failed) — tools/gpu_scale.sh fails on rccl_ranks != N
               "rccl_ranks": int(rccl_probe["rccl_ranks"]), "rccl_probe": (rccl_probe["when"] or ("not run: " + ("rehearsal over gloo" if rehearsal else "single process" if not use_dist else "RCCL group never came up"))),
               **({"rehearsal": "all ranks on one GPU over gloo: a functional run of the multi-rank paths, not a measurement"} if rehearsal else {}), "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(dt / a.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "strong" if tiled else "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": WORKLOADS[a.config] + f", {intr.width}x{intr.height} orbit stream 0.5 deg/frame, all {pixels} pixels integrated and meshed "
                                                            "every frame, resolution 4, fast decode, max_std 0.15",
                          "mode": a.mode, "points_per_frame": pixels,
                          "rccl_before_clock": (bool(use_dist and not clock_over_gloo) if use_dist else None),
                          "mlp_pipe": ("bf16x6: every fp32 product of the MLP tiles as six exact bf16 slice products on v_mfma_f32_32x32x16_bf16, fp32 accumulate "
                                       "(fp32-equivalent: same parity bars as the f32-input MFMA kernels, which DIF_DECODER_PIPE=f32 selects)" if pipe == "bf16x6"
                                       else "f32: v_mfma_f32_32x32x2_f32"),
                          "parallelism": (f"slab {a.loopback // 2} of {a.loopback} x-slabs of one stream on ONE GPU, halo exchange with itself (loopback) after every integrate"
                                          if a.loopback > 1 else
                                          f"one stream, grid cut into {world} x-slabs, halo exchange (RCCL send/recv, 3 boundary layers) after every integrate"
                                          if tiled else f"{world * S_main} independent subsequences, {S_main} private maps per GPU sharing their launches" if gb is not None
                                          else f"{world} independent subsequences (one map per GPU)"),
                          "streams_per_gpu": max(S_main, 1),
                          "d2h_per_frame": a.d2h,
                          "d2h_engine": (None if a.d2h != "dma" else "sdma (hsa_amd_memory_async_copy, no copy kernel)" if stream.sdma else
                                         "hipMemcpyAsync on a side stream (blit kernels)"),
                          "sdma_call_us": (None if not stream.sdma_us else {"calls": len(stream.sdma_us), "median": round(float(np.median([u for _, u in stream.sdma_us])), 1),
                                                                               "p90": round(float(np.percentile([u for _, u in stream.sdma_us], 90)), 1),
                                                                               "first_40": [(n, round(u)) for n, u in stream.sdma_us[:40]]}),
                          "two_queues": {"on": bool(stream.overlap), "queues_independent": stream.queues_independent,
                                         "what": "frame i+1's integrate front end on a second hardware queue beside frame i's extract; fusion kernel and extract "
                                                 "ordered by hipStreamWaitValue32 on words the kernels publish" if stream.overlap else "one queue"},
                          "host_pipeline_depth": 2 if a.pipeline else 1, "launch": launch,
                          "avg_per_frame_rank0": {k: round(float(np.mean([s[k] for s in st])), 1)
                                                  for k in ("M", "C", "K", "B", "VH", "T", "n_occupied", "cache_T")},
                          "frames_per_s_with_mesh_left_in_hbm": hbm_resident,
                          "frames_per_s_with_S_streams_per_gpu": ({str(S): v[0] for S, v in by_streams.items()} if by_streams else None),
                          "mesh_log_compactions": stream.map._gc_epoch,
                          "extract_buffers": {"rows": int(stream.map._xbuf[0][1]) if stream.map._xbuf else None, "deferred_extracts": int(stream.map.n_deferred),
                                              "bytes": int(sum(t.numel() * t.element_size() for t in stream.map._xbuf[1].values())) if stream.map._xbuf else None},
                          "global_map_merge_after_the_clock": merge_info,
                          "halo_exchange": halo_summary(stream, a)},
               "roofline": roofline_block(per_frame, [st[j] for j in timed_idx], pipe, short_run=(a.steps <= 30), streams=max(S_main, 1))}
        if by_streams and out["roofline"] is not None:
            out["roofline"]["by_streams"] = {str(S): v[1] for S, v in by_streams.items()}
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(stream, a.config, scene, cfg, intr, n_frames, a.cpu_frames)
    if use_dist:
        flush_c_stdio()
        # every rank has flushed whatever it had to say before rank 0 prints the one JSON line — unless the epilogue (here or on another rank)
        # is stuck in a collective: then the line goes out without the barrier and the process leaves without tearing the group down
        if epilogue_ok:
            epilogue_ok, _ = within(60.0, lambda: (dist.barrier(), dist.destroy_process_group()), "final barrier", dev)
    flush_c_stdio()
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if use_dist and not epilogue_ok:
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
