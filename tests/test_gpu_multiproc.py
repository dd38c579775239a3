"""GPU: the multi-process legs of BASELINE configs C4 and C5 with TWO LIVE PROCESSES and the real kernels on ONE GPU.

RCCL refuses two ranks on one device, so the process group is gloo and `di_fusion_amd.parallel` stages the messages through pinned host
memory (the transport switch in `all_gather_records` / `HaloExchange.transfer`); everything else — `FusionStream(tiling=...)`, the
delta / whole-layer halo protocol, `build_global_map`, the export / merge kernels — is exactly what runs over RCCL on an 8-GPU node.
What is left untested on a one-GPU box is the RCCL transport itself (`tests/test_gpu_parallel.py::test_spatial_tiling_two_processes_rccl`
runs where two GPUs are visible)."""
import os
import socket

import numpy as np
import pytest
import torch

from di_fusion_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")

# (scene, map config, intrinsic scale, frames, degrees per frame)
STREAMS = {
    "room16": (syn.default_room, lambda: syn.MapConfig((-3.2,) * 3, (3.2,) * 3, 0.4), 0.25, 4, 15.0),
    # BASELINE config C5: ONE 1280x960 stream on the C3 grid (128^3): frames 0-1 travel as whole layers, later ones as bounded deltas
    "c5_1280x960": (lambda: syn.config_c3()[0], lambda: syn.config_c3()[1], 2.0, 6, 0.5),
}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _init(rank, world, port):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(DEV)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def _model():
    from di_fusion_amd.network import utility as net_util
    return net_util.networks_from_arrays(net_util.load_weights_npz())


def _tiled_worker(rank, world, port, case, q):
    dist = _init(rank, world, port)
    from di_fusion_amd.stream import FusionStream
    scene_f, cfg_f, scale, n_frames, deg = STREAMS[case]
    st = FusionStream(_model(), scene_f(), cfg_f(), syn.Intrinsic().scaled(scale), DEV, n_frames, deg_per_frame=deg, tiling=(rank, world, None),
                      initial_capacity=1 << 14)
    tris = []
    for i in range(n_frames):
        out = st.step(i, d2h="new")
        torch.cuda.synchronize()                                # (the hand-over to pinned memory is asynchronous)
        tris.append(None if out is None else tuple(x.numpy().copy() for x in out))
    m = st.map
    n = m.n_occupied
    kinds = [st._halo_buffers["hist"][f]["kinds"] for f in sorted(st._halo_buffers["hist"])]
    q.put((rank, m._ownership, m.latent_vecs_pos[:n].cpu().numpy(), m.voxel_obs_count[:n].cpu().numpy(), m.latent_vecs[:n].cpu().numpy(), tris, kinds))
    dist.barrier()
    dist.destroy_process_group()


def _spawn(target, args_of_rank, world=2, timeout=900):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=args_of_rank(r) + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=timeout) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("case", list(STREAMS))
def test_spatial_tiling_two_processes_one_gpu(case, gpu_model):
    """C5 with two live processes: `FusionStream(tiling=(rank, 2, ...))`, halo exchange through the process group after every integrate.
    Owned voxels of both ranks equal the single stream bit for bit after the last frame, and so does every frame's mesh update (the
    union of the two slabs' new triangles)."""
    from di_fusion_amd.stream import FusionStream
    port = _free_port()
    res = _spawn(_tiled_worker, lambda r: (r, 2, port, case))
    scene_f, cfg_f, scale, n_frames, deg = STREAMS[case]
    full = FusionStream(gpu_model, scene_f(), cfg_f(), syn.Intrinsic().scaled(scale), DEV, n_frames, deg_per_frame=deg, initial_capacity=1 << 14)
    full_tris = []
    for i in range(n_frames):
        out = full.step(i, d2h="new")
        torch.cuda.synchronize()
        full_tris.append(None if out is None else tuple(x.numpy().copy() for x in out))
    nF = full.map.n_occupied
    posF = full.map.latent_vecs_pos[:nF].cpu().numpy()
    wF, zF = full.map.voxel_obs_count[:nF].cpu().numpy(), full.map.latent_vecs[:nF].cpu().numpy()
    plane = full.map.n_xyz[1] * full.map.n_xyz[2]
    covered = 0
    for rank, (lo, hi, _), pos, w, z, _, _ in res:
        own_f, own_s = (posF >= lo * plane) & (posF < hi * plane), (pos >= lo * plane) & (pos < hi * plane)
        of, os_ = np.argsort(posF[own_f]), np.argsort(pos[own_s])
        assert np.array_equal(pos[own_s][os_], posF[own_f][of])
        assert np.array_equal(w[own_s][os_], wF[own_f][of]) and np.array_equal(z[own_s][os_], zF[own_f][of])
        covered += own_f.sum()
    assert covered == nF
    for f in range(n_frames):                                   # every frame's mesh update, bit for bit
        parts = [r[5][f] for r in res if r[5][f] is not None]
        if full_tris[f] is None:
            assert not parts
            continue
        tS, iS = np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
        tF, iF = full_tris[f][0], full_tris[f][1]
        assert tS.shape == tF.shape, (f, tS.shape, tF.shape)
        kS = np.lexsort(tuple(tS.reshape(len(tS), -1).T[::-1]) + (iS,))
        kF = np.lexsort(tuple(tF.reshape(len(tF), -1).T[::-1]) + (iF,))
        assert np.array_equal(iS[kS], iF[kF]) and np.array_equal(tS[kS], tF[kF])
    if case == "c5_1280x960":                                   # the bounded delta messages really were used, by both ends alike
        k0, k1 = res[0][6], res[1][6]
        assert any(ko["right"] == "delta" for ko, _ in k0) and any(ki["left"] == "delta" for _, ki in k1)
        for (ko0, ki0), (ko1, ki1) in zip(k0, k1):
            assert ko0["right"] == ki1["left"] and ko1["left"] == ki0["right"]      # both ends chose the same kind without talking


def _merge_worker(rank, world, port, q):
    dist = _init(rank, world, port)
    from di_fusion_amd import parallel
    from di_fusion_amd.stream import FusionStream
    from di_fusion_amd.system.map import DenseIndexedMap
    scene, cfg = syn.config_c2()
    model = _model()
    st = FusionStream(model, scene, cfg, syn.Intrinsic(), DEV, 3, deg_per_frame=2.0, phase_deg=45.0 * rank, initial_capacity=1 << 14)
    for i in range(3):
        st.step(i, d2h="none")
    g = parallel.build_global_map(st.map, lambda: DenseIndexedMap(model, cfg.namespace(), 29, DEV, initial_capacity=1 << 14))
    n = g.n_occupied
    q.put((rank, g.latent_vecs_pos[:n].cpu().numpy(), g.voxel_obs_count[:n].cpu().numpy(), g.latent_vecs[:n].cpu().numpy(), st.map.n_occupied))
    dist.barrier()
    dist.destroy_process_group()


def test_global_map_merge_two_processes_one_gpu(gpu_model):
    """C4 with two live processes: each fuses its own arc of the orbit (C2 grid, full 640x480 frames), then `build_global_map` —
    the variable-length all-gather of voxel records and the fold in rank order.  Both ranks end with the same map, bit for bit, and it
    equals the fold of the two maps built in this process."""
    from di_fusion_amd.stream import FusionStream
    from di_fusion_amd.system.map import DenseIndexedMap
    port = _free_port()
    res = _spawn(_merge_worker, lambda r: (r, 2, port))
    scene, cfg = syn.config_c2()
    g = DenseIndexedMap(gpu_model, cfg.namespace(), 29, DEV, initial_capacity=1 << 14)
    local_n = []
    for rank in range(2):
        st = FusionStream(gpu_model, scene, cfg, syn.Intrinsic(), DEV, 3, deg_per_frame=2.0, phase_deg=45.0 * rank, initial_capacity=1 << 14)
        for i in range(3):
            st.step(i, d2h="none")
        local_n.append(st.map.n_occupied)
        g.merge_records(st.map.export_records())
    n = g.n_occupied
    want = (g.latent_vecs_pos[:n].cpu().numpy(), g.voxel_obs_count[:n].cpu().numpy(), g.latent_vecs[:n].cpu().numpy())
    assert n > max(local_n)
    for rank, pos, w, z, n_local in res:
        assert n_local == local_n[rank]
        assert np.array_equal(pos, want[0]) and np.array_equal(w, want[1]) and np.array_equal(z, want[2])


def test_merge_of_eight_c3_maps_vs_oracle(gpu_model, oracle_net):
    """BASELINE config C4's merge at its real size: eight maps of the C3 stream (128^3 grid, 0.05 m, full 640x480 frames; arc r starts
    at r * 45 degrees of the bench orbit), folded in rank order by `dif_merge_records`, against the oracle's `merge_records` on the same
    records (reference arithmetic: /root/reference/pytorch/system/map.py:448-451): slot order, observation counts and dirty set bit-exact,
    latents to 1e-5."""
    from di_fusion_amd.stream import FusionStream
    from di_fusion_amd.system.map import DenseIndexedMap
    from oracle import difusion_oracle as O
    scene, cfg = syn.config_c3()
    recs = []
    for rank in range(8):
        st = FusionStream(gpu_model, scene, cfg, syn.Intrinsic(), DEV, 4, deg_per_frame=0.5, phase_deg=45.0 * rank, initial_capacity=1 << 16)
        for i in range(4):
            st.step(i, d2h="none")
        recs.append(st.map.export_records())
        del st
    sizes = [r.size(0) for r in recs]
    assert min(sizes) > 10_000, sizes                             # C3-size maps: tens of thousands of voxels each
    g = DenseIndexedMap(gpu_model, cfg.namespace(), 29, DEV, initial_capacity=1 << 16)
    go = O.OracleMap(oracle_net, cfg.bound_min, cfg.bound_max, cfg.voxel_size)
    for r in recs:
        g.merge_records(r)
        O.merge_records(go, r.cpu().numpy())
    n = go.n_occupied
    assert g.n_occupied == n and n > max(sizes)
    assert np.array_equal(g.latent_vecs_pos[:n].cpu().numpy(), go.latent_vecs_pos[:n])
    assert np.array_equal(g.voxel_obs_count[:n].cpu().numpy(), go.voxel_obs_count[:n])
    assert np.abs(g.latent_vecs[:n].cpu().numpy() - go.latent_vecs[:n]).max() <= 1e-5
    assert np.array_equal(g.updated_vec_id.cpu().numpy(), np.sort(go.updated_vec_id))
    # overlap exists (neighbouring arcs see the same walls), so the fold really added weights
    assert n < sum(sizes)
    v = g.extract_mesh_arrays(4, int(8e6), max_std=0.15, to_host=False)
    assert v is not None and v[0].shape[0] > 10_000


@pytest.mark.parametrize("mode", ["c4", "c4s2", "tiled"])
def test_bench_multi_rank_paths_rehearsal(mode):
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one process per rank), both ranks on this box's one
    GPU over gloo (`DIF_BENCH_REHEARSAL=1`): barrier, max-over-ranks timing, the JSON contract, and — c4 — the all-gather merge of the
    two maps with the global mesh / — tiled — one 1280x960 stream over two slabs with the per-frame halo exchange.  Numbers mean nothing
    here; a crash or a hang in these paths would first show up on the 8-GPU node otherwise."""
    import json
    import subprocess
    import sys
    from tests.conftest import ROOT
    env = dict(os.environ, DIF_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "3", "--no-cpu-baseline",
           "--mode", "tiled" if mode == "tiled" else "c4"] + (["--streams-per-gpu", "2"] if mode == "c4s2" else [])       # c4s2: two streams per rank, batched launches
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    line = [l for l in p.stdout.strip().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 3 and d["value"] > 0 and d["higher_is_better"] is True
    assert d["scaling"] == ("strong" if mode == "tiled" else "weak") and "rehearsal" in d
    if mode != "tiled":
        m = d["config"]["global_map_merge_after_the_clock"]
        assert "error" not in m, m
        assert m["global_voxels"] > 0 and m["local_voxels_rank0"] > 0 and m["global_mesh_triangles"] > 0
        assert d["config"]["streams_per_gpu"] == (2 if mode == "c4s2" else 1)
        if mode == "c4":
            assert m["global_voxels"] > m["local_voxels_rank0"]
    else:
        h = d["config"]["halo_exchange"]
        assert h["mode"] == "delta" and h["bytes_sent_per_frame"] > 0
