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
