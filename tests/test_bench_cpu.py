"""CPU: the parts of bench.py that do not need a GPU — the roofline arithmetic and the launcher contract."""
import os
import subprocess
import sys

from tests.conftest import ROOT


def test_roofline_block_arithmetic():
    sys.path.insert(0, str(ROOT))
    import bench
    # two event-timed frames: rows from the counters, milliseconds from the events
    sst = [dict(M=64000, B=1000, VH=32000), dict(M=32000, B=500, VH=16000)]
    per_frame = [[("encode", 0.050), ("decode_lattice", 0.060), ("decode_points", 0.040), ("mc_count", 0.017), ("mc_emit", 0.013)],
                 [("encode", 0.030), ("decode_lattice", 0.040), ("decode_points", 0.020), ("mc_count", 0.015), ("mc_emit", 0.011)]]
    want = (1000 + 500) * 64 * bench.DEC_FLOP_PER_ROW / ((0.060 + 0.040) * 1e-3) / 1e12
    enc = (64000 + 32000) * bench.ENC_FLOP_PER_ROW / (0.080e-3) / 1e12
    for pipe, peak in (("f32", bench.PEAK_FP32_MFMA_TFLOPS), ("bf16x6", bench.PEAK_BF16_MFMA_TFLOPS / 6)):
        r = bench.roofline_block(per_frame, sst, pipe)
        assert r["kernel"] == "k_decode_voxels" and r["bound"] == "mfma" and r["event_timed_frames"] == 2
        # `achieved` is the reference's algorithmic fp32 FLOP rate whatever pipe runs it; `peak` is that pipe's ceiling for it
        assert abs(r["achieved"] - want) < 1e-2 and abs(r["frac"] - want / peak) < 1e-3 and abs(r["peak"] - peak) < 0.1
        assert abs(r["frac_of_f32_input_mfma_peak"] - want / bench.PEAK_FP32_MFMA_TFLOPS) < 1e-3
        assert abs(r["per_kernel"]["encode"]["tflops"] - enc) < 1e-2
        first, second = r["by_phase"]["first_half_of_timed_frames"], r["by_phase"]["second_half_of_timed_frames"]
        assert abs(first["decode_lattice"]["frac"] - 1000 * 64 * bench.DEC_FLOP_PER_ROW / 0.060e-3 / 1e12 / peak) < 1e-3
        assert abs(second["encode"]["avg_launch_ms"] - 0.030) < 1e-6
        assert abs(r["other_ms_per_frame"]["mc_count"] - 0.016) < 1e-6
        assert r["traffic_source"] is None or "not measured by this run" in r["traffic_source"]
        # matrix-pipe busy fraction: the tiles' MFMA cycles over (1,024 SIMDs x the launch)
        cyc = bench.TILE_PIPE_CYCLES[pipe]["decode_lattice"]
        busy = (1500 * 64 / 2 / 32) * cyc / (1024 * 2.4e9 * 0.050e-3)
        assert abs(r["pipe_busy_frac"] - busy) < 1e-3 and 0 < r["pipe_busy_frac"] < 1
    # on the bf16 pipe the same arithmetic costs 6/16 of the f32-input MFMA's pipe time: the ceiling is 2,500 / 6 TFLOP/s
    assert bench.TILE_PIPE_CYCLES["bf16x6"]["decode_lattice"] == 480 * 32 + 16 * 64 and abs(bench.PEAK_X6_FP32_EQUIV_TFLOPS - 416.67) < 0.01


def _run(args, env_extra):
    env = dict(os.environ, **env_extra)
    env.pop("RANK", None)
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_launcher_contract_is_loud():
    """--gpus must agree with the ranks that really exist: a WORLD_SIZE mismatch and a request for more GPUs than are visible both fail
    with a message instead of silently running one rank."""
    p = _run(["--gpus", "1"], {"WORLD_SIZE": "2"})
    assert p.returncode != 0 and "WORLD_SIZE=2" in (p.stderr + p.stdout)
    import torch
    n = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k != "WORLD_SIZE"}
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(n + 2)], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "GPU(s) visible" in (p.stderr + p.stdout)


def _probe_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, str(ROOT))
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    whole = bench.collective_probe(None, torch.device("cpu"))
    alone = bench.collective_probe(None, torch.device("cpu"))      # (a second probe on the same group)
    q.put((rank, whole, alone))
    dist.barrier()
    dist.destroy_process_group()


def test_collective_probe_counts_the_ranks_it_really_reaches():
    """`rccl_ranks` of the bench line is OBSERVED: the world size of the group after an all-reduce over it returned the right sum.  Two
    live gloo ranks -> 2 on both; a group whose collective fails, or returns a wrong sum, -> 0."""
    import socket

    import torch
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_probe_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, 2, 2), (1, 2, 2)]
    # no process group at all (the RCCL group "never came up"): the probe reports 0 instead of raising
    sys.path.insert(0, str(ROOT))
    import bench
    assert bench.collective_probe(None, torch.device("cpu")) == 0

    class Silent:           # a group whose all-reduce returns without summing (what a half-connected transport would look like)
        pass
    import torch.distributed as dist
    orig = (dist.get_world_size, dist.get_rank, dist.all_reduce)
    try:
        dist.get_world_size, dist.get_rank, dist.all_reduce = (lambda g=None: 2), (lambda g=None: 0), (lambda t, op=None, group=None: None)
        assert bench.collective_probe(Silent(), torch.device("cpu")) == 0
    finally:
        dist.get_world_size, dist.get_rank, dist.all_reduce = orig


def test_every_json_under_profiles_parses():
    """profiles/*.json are read by bench.py (static PMC figures) and cited by the judge: none may carry a launcher banner in front of its line
    (tools/install_profiles.sh strips them)."""
    import json
    bad = []
    for p in sorted((ROOT / "profiles").glob("*.json")):
        try:
            json.loads(p.read_text())
        except Exception as e:
            bad.append((p.name, repr(e)[:80]))
    for p in sorted((ROOT / "profiles").glob("*.jsonl")):
        for k, line in enumerate(p.read_text().strip().splitlines()):
            try:
                json.loads(line)
            except Exception as e:
                bad.append((f"{p.name}:{k}", repr(e)[:80]))
    assert not bad, bad


def _scale_line(n, value, leg, **over):
    tiled = leg == "tiled"
    d = {"metric": "frames/s", "value": value, "n_gpus": n, "rccl_ranks": (n if n > 1 else 0), "rccl_probe": "test", "ms_per_step": 0.1,
         "scaling": "strong" if tiled else "weak",
         "config": {"rccl_before_clock": (leg in ("c4rccl", "tiled")) if n > 1 else None,
                    "global_map_merge_after_the_clock": None if tiled else {"all_gather_and_fold_ms": 1.0}}}
    d.update(over)
    return d


def test_scale_tool_dry_run_and_checks(tmp_path, capsys):
    """tools/gpu_scale.py, the command of the 8-GPU day, without a GPU: `--dry` lists every launch line (all four legs x N = 1, 2, 4, 8, RCCL on
    both sides of the clock, 127.0.0.1 rendezvous, one rank per GPU) and the predicted bands; `check()` accepts a curve inside the bands and fails on
    an unobserved RCCL group, a failed map merge, RCCL on the wrong side of the clock, an N = 1 line that disagrees with the plain bench, and a curve
    outside the prediction of DESIGN.md section 6."""
    import json
    sys.path.insert(0, str(ROOT / "tools"))
    import gpu_scale
    assert gpu_scale.main(["--dry", "--out", str(tmp_path)]) == 0
    text = capsys.readouterr().out
    for leg in ("c4", "c4rccl", "c4s4", "tiled"):
        for n in (1, 2, 4, 8):
            assert f"[{leg} N={n}]" in text
    assert text.count("--rccl-before-clock 1") == 4 and text.count("--master-addr 127.0.0.1") == 12 and "HSA_ENABLE_IPC_MODE_LEGACY=0" in text
    assert "rank 7: RANK=7 LOCAL_RANK=7 WORLD_SIZE=8 MASTER_ADDR=127.0.0.1" in text and "predicted bands" in text
    cmds = gpu_scale.commands(tmp_path, 200, 8)
    assert len(cmds) == 17 and len({str(c[3]) for c in cmds}) == 17
    ports = [c[2][c[2].index("--master-port") + 1] for c in cmds if "--master-port" in c[2]]
    assert len(set(ports)) == len(ports) == 12                                    # no two launches share a rendezvous port

    def write(lines):
        for f in tmp_path.glob("*.json"):
            f.unlink()
        (tmp_path / "ref_n1.json").write_text("banner\n" + json.dumps({"value": 7000.0}))
        for (leg, n), d in lines.items():
            (tmp_path / f"{leg}_n{n}.json").write_text(json.dumps(d))

    good = {}
    for leg in ("c4", "c4rccl", "c4s4"):
        for n in (1, 2, 4, 8):
            good[(leg, n)] = _scale_line(n, 7000.0 * n * (0.97 if n > 1 else 1.0) * (1.6 if leg == "c4s4" else 1.0), leg)
    for n, sp in ((1, 1.0), (2, 1.1), (4, 1.25), (8, 1.33)):
        good[("tiled", n)] = _scale_line(n, 5000.0 * sp, "tiled")
    write(good)
    out = []
    assert gpu_scale.check(tmp_path, out.append) and out[-1] == "CHECKS ok"
    for what, key, over in (("rccl_ranks=0", ("c4", 4), dict(rccl_ranks=0)),
                            ("global map merge failed", ("c4", 2), dict(config={"rccl_before_clock": False, "global_map_merge_after_the_clock": {"error": "x"}})),
                            ("rccl_before_clock=False, wanted True", ("c4rccl", 2), dict(config={"rccl_before_clock": False, "global_map_merge_after_the_clock": {}})),
                            ("differs from the plain bench", ("c4", 1), dict(value=6000.0)),
                            ("outside the predicted band", ("c4", 8), dict(value=7000.0 * 8 * 0.7)),
                            ("outside the predicted band", ("tiled", 8), dict(value=5000.0 * 2.5))):
        bad = dict(good)
        bad[key] = {**good[key], **over}
        write(bad)
        out = []
        assert not gpu_scale.check(tmp_path, out.append), what
        assert any(what in ln for ln in out), (what, out)


def test_scale_script_fails_on_unobserved_rccl():
    """tools/gpu_scale.py checks the OBSERVED rank count, the merge outcome and which side of the clock RCCL came up on."""
    t = (ROOT / "tools" / "gpu_scale.py").read_text()
    assert "rccl_ranks" in t and "rccl_probe" in t and "global map merge failed" in t and "rccl_before_clock" in t and "c4rccl" in t
    assert "gpu_scale.py" in (ROOT / "tools" / "gpu_scale.sh").read_text()
