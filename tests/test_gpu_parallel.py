"""GPU: record export / merge kernels against the oracle (single process; the collective itself is covered by gloo on CPU)."""
import numpy as np
import pytest
import torch

from di_fusion_amd import synthetic as syn
from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def test_export_merge_records_vs_oracle(gpu_model, oracle_net):
    from di_fusion_amd.system.map import DenseIndexedMap
    from oracle import difusion_oracle as O
    cfg = syn.MapConfig((-1.6,) * 3, (1.6,) * 3, 0.4)
    intr = syn.Intrinsic().scaled(0.125)
    scene = syn.Scene(kind="sphere", radius=1.3)
    gm, om = [], []
    for rank in range(2):
        m = DenseIndexedMap(gpu_model, cfg.namespace(), 29, DEV, initial_capacity=1024)
        o = O.OracleMap(oracle_net, cfg.bound_min, cfg.bound_max, cfg.voxel_size)
        for f in range(2):
            xyz, nrm = syn.frame_points(scene, f, intr, deg_per_frame=20.0, phase_deg=90.0 * rank)
            m.integrate_keyframe(xyz.to(DEV), nrm.to(DEV))
            o.integrate_keyframe(xyz.numpy(), nrm.numpy())
        gm.append(m); om.append(o)
    recs = [m.export_records() for m in gm]
    for r, o in zip(recs, om):
        want = O.export_records(o)
        got = r.cpu().numpy()
        assert np.array_equal(got[:, :3], want[:, :3])                     # lin ids and weights bit-exact
        assert np.abs(got[:, 3:].view(np.float32) - want[:, 3:].view(np.float32)).max() < 1e-3
    g = DenseIndexedMap(gpu_model, cfg.namespace(), 29, DEV, initial_capacity=1024)
    go = O.OracleMap(oracle_net, cfg.bound_min, cfg.bound_max, cfg.voxel_size)
    for r in recs:
        g.merge_records(r)
        O.merge_records(go, r.cpu().numpy())
    n = go.n_occupied
    assert g.n_occupied == n
    assert np.array_equal(g.latent_vecs_pos[:n].cpu().numpy(), go.latent_vecs_pos[:n])
    assert np.array_equal(g.voxel_obs_count[:n].cpu().numpy(), go.voxel_obs_count[:n])
    assert np.abs(g.latent_vecs[:n].cpu().numpy() - go.latent_vecs[:n]).max() < 1e-5
    assert np.array_equal(g.updated_vec_id.cpu().numpy(), go.updated_vec_id)
    # the merged map meshes
    v, vid, vs = g.extract_mesh_arrays(4, int(4e6), max_std=0.15)
    assert v.shape[0] > 0


TILING_CASES = {
    # 16^3 grid, quarter-resolution frames, large yaw: allocation, the 600-count gate and re-meshing all happen across the cut
    "room16_2slabs": (syn.default_room(), syn.MapConfig((-3.2,) * 3, (3.2,) * 3, 0.4), 0.25, 2, 4, 15.0, "full"),
    # BASELINE config C5: ONE 1280x960 stream (1.23 M points per frame) on the C3 grid (128^3, 0.05 m), cut into 2 and into 8 x-slabs:
    # whole-layer messages every frame ...
    "c5_1280x960_2slabs": (*syn.config_c3(), 2.0, 2, 2, 0.5, "full"),
    "c5_1280x960_8slabs": (*syn.config_c3(), 2.0, 8, 2, 0.5, "full"),
    # ... and the bounded delta protocol (frames 0-1 whole layers, then only what changed, at most 4,096 records per message)
    "c5_1280x960_2slabs_delta": (*syn.config_c3(), 2.0, 2, 6, 0.5, "delta"),
    "c5_1280x960_8slabs_delta": (*syn.config_c3(), 2.0, 8, 6, 0.5, "delta"),
}


@pytest.mark.parametrize("case", list(TILING_CASES))
def test_spatial_tiling_matches_single_map_bit_for_bit(case, gpu_model):
    """C5: x-slabs with halo exchange (emulated in one process: `parallel.HaloExchange` phase by phase over all slabs, the same kernels,
    message kinds and record path as the RCCL version, messages handed over by device copies) reproduce the single-map state of every
    owned voxel BIT FOR BIT, and the union of the slab meshes equals the single-map mesh."""
    from di_fusion_amd import parallel
    from di_fusion_amd.system.map import DenseIndexedMap
    scene, cfg, scale, world, n_frames, deg, mode = TILING_CASES[case]
    intr = syn.Intrinsic().scaled(scale)
    full = DenseIndexedMap(gpu_model, cfg.namespace(), 29, DEV, initial_capacity=2048)
    nx = full.n_xyz[0]
    slabs, states = [], []
    for r in range(world):
        m = DenseIndexedMap(gpu_model, cfg.namespace(), 29, DEV, initial_capacity=2048)
        m.set_ownership(*parallel.slab_range(nx, r, world), halo=parallel.HALO)
        slabs.append(m)
        states.append({})
    plane = full.n_xyz[1] * full.n_xyz[2]
    delta_messages = delta_bytes = full_bytes = delta_records = 0
    for f in range(n_frames):
        xyz, nrm = syn.frame_points(scene, f, intr, deg_per_frame=deg)
        xyz, nrm = xyz.to(DEV), nrm.to(DEV)
        assert scale != 2.0 or xyz.size(0) > 1_100_000         # a 1280x960 frame
        full.integrate_keyframe(xyz, nrm)
        for m in slabs:
            m.integrate_keyframe(xyz, nrm)
        # halo exchange: every rank exports, the messages change hands (what RCCL send/recv does), every rank merges
        xs = [parallel.HaloExchange(m, r, world, states[r], mode) for r, m in enumerate(slabs)]
        for x in xs:
            x.export()
        for r, x in enumerate(xs):
            for name, peer, _, _ in x.sides:
                other = "right" if name == "left" else "left"
                assert x.n_in[name] == xs[peer].n_out[other]          # both ends chose the same message kind without talking
                x.inp[name][:x.n_in[name]].copy_(xs[peer].out[other][:x.n_in[name]])
                if x.kind_in[name] == "delta":
                    delta_messages += 1
                    delta_bytes += x.n_in[name] * 128
                else:
                    full_bytes += x.n_in[name] * 128
        for x in xs:
            x.merge()
        torch.cuda.synchronize()
        for x in xs:
            for name, _, k, _ in x.sides:
                if x.kind_out[name] == "delta":
                    n_rec, pending = int(x.note_out[4 * k]), int(x.note_out[4 * k + 1])
                    assert n_rec == pending <= parallel.DELTA_ROWS      # nothing was cut off
                    delta_records += n_rec
        # ---- state of owned voxels ----
        nF = full.n_occupied
        posF = full.latent_vecs_pos[:nF].cpu().numpy()
        wF = full.voxel_obs_count[:nF].cpu().numpy()
        zF = full.latent_vecs[:nF].cpu().numpy()
        covered = 0
        for r, m in enumerate(slabs):
            lo, hi = m._ownership[0] * plane, m._ownership[1] * plane
            own = (posF >= lo) & (posF < hi)
            idx = m.indexer.cpu().numpy()[posF[own]]
            assert (idx >= 0).all(), f"frame {f} rank {r}: owned voxel not allocated"
            assert np.array_equal(m._obs.cpu().numpy()[idx], wF[own])
            assert np.array_equal(m._latent.cpu().numpy()[idx], zF[own]), f"frame {f} rank {r}: latent bits differ"
            # and nothing extra is allocated inside the owned slab
            nS = m.n_occupied
            posS = m.latent_vecs_pos[:nS].cpu().numpy()
            assert ((posS >= lo) & (posS < hi)).sum() == own.sum()
            covered += own.sum()
            # the halo layers mirror their owners exactly: same voxels, same (w, z) bits
            for side, nb in (("left", r - 1), ("right", r + 1)):
                if not 0 <= nb < world:
                    continue
                h_lo, h_hi = (m._ownership[0] - parallel.HALO, m._ownership[0]) if side == "left" else (m._ownership[1], m._ownership[1] + parallel.HALO)
                inh = (posF >= h_lo * plane) & (posF < h_hi * plane)
                hidx = m.indexer.cpu().numpy()[posF[inh]]
                assert (hidx >= 0).all(), f"frame {f} rank {r}: halo voxel missing"
                assert np.array_equal(m._obs.cpu().numpy()[hidx], wF[inh]) and np.array_equal(m._latent.cpu().numpy()[hidx], zF[inh])
        assert covered == nF
        # ---- meshes ----
        vF, idF, _ = full.extract_mesh_arrays(4, int(4e6), max_std=0.15)
        newF = full.mesh_cache_tensors(new_only=True)
        tris, ids = [], []
        for m in slabs:
            m.extract_mesh_arrays(4, int(4e6), max_std=0.15)
            new = m.mesh_cache_tensors(new_only=True)
            if new is not None:                                    # a slab the camera has not reached yet has no mesh
                tris.append(new[0].cpu().numpy()); ids.append(new[1].cpu().numpy())
        tS, iS = np.concatenate(tris), np.concatenate(ids)
        tF, iF = newF[0].cpu().numpy(), newF[1].cpu().numpy()
        assert tS.shape == tF.shape, (f, tS.shape, tF.shape)
        kS = np.lexsort(tuple(tS.reshape(len(tS), -1).T[::-1]) + (iS,))
        kF = np.lexsort(tuple(tF.reshape(len(tF), -1).T[::-1]) + (iF,))
        assert np.array_equal(iS[kS], iF[kF])
        assert np.array_equal(tS[kS], tF[kF])                  # bit-identical vertices
    if mode == "delta":
        assert delta_messages > 0 and delta_records > 0, "the stream never reached the bounded delta messages"
        print(f"{case}: {delta_messages} delta messages ({delta_bytes} B, {delta_records} records), whole-layer messages {full_bytes} B")


def test_tiled_direct_launches_match_eager_steps(gpu_model):
    """The spatially tiled stream driven like the headline stream (`step_direct`: two C calls per frame with the halo refresh enqueued
    between them, host one frame ahead) against eager `step`s, in the loopback arrangement `bench.py --mode tiled --loopback 8` times
    (slab 4 of 8 exchanging with itself): same map, bit for bit, same mesh updates."""
    from di_fusion_amd.stream import FusionStream
    scene, cfg = syn.config_c3()
    outs, maps = {}, {}
    for how in ("eager", "direct"):
        st = FusionStream(gpu_model, scene, cfg, syn.Intrinsic(), DEV, 8, deg_per_frame=0.5, tiling=(4, 8, None), halo_loopback=True,
                          initial_capacity=1 << 16)
        res = []
        for i in range(8):
            if how == "eager" or i < 2:
                o = st.step(i, "new")
                torch.cuda.synchronize()
                res.append(None if o is None else tuple(x.numpy().copy() for x in o))
            else:
                o = st.step_direct(i, "new")
                if o is not None or i > 2:
                    res.append(None if o is None else tuple(x.numpy().copy() for x in o))
        o = st.flush("new")
        if how == "direct":
            res.append(None if o is None else tuple(x.numpy().copy() for x in o))
        n = st.map.n_occupied
        maps[how] = (st.map.latent_vecs_pos[:n].cpu().numpy(), st.map.voxel_obs_count[:n].cpu().numpy(), st.map.latent_vecs[:n].cpu().numpy())
        outs[how] = res
        kinds = [st._halo_buffers["hist"][f]["kinds"][0] for f in sorted(st._halo_buffers["hist"])]
        assert any("delta" in k.values() for k in kinds)
    for a, b in zip(maps["eager"], maps["direct"]):
        assert np.array_equal(a, b)
    assert len(outs["eager"]) == len(outs["direct"]) == 8
    for a, b in zip(outs["eager"], outs["direct"]):
        assert (a is None) == (b is None)
        if a is not None:
            assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_delta_halo_overflow_is_reported(gpu_model):
    """A delta message that cannot hold the frame's changes is reported, not silently truncated: the device flags it (DIF_C_OVERFLOW = 8,
    raised by the next counter read) and the header carries pending > records."""
    from di_fusion_amd import parallel
    from di_fusion_amd.system.map import DenseIndexedMap
    scene, cfg = syn.config_c3()
    m = DenseIndexedMap(gpu_model, cfg.namespace(), 29, DEV, initial_capacity=1 << 16)
    m.set_ownership(*parallel.slab_range(m.n_xyz[0], 3, 8), halo=parallel.HALO)
    xyz, nrm = syn.frame_points(scene, 0, syn.Intrinsic(), deg_per_frame=0.5)
    m.integrate_keyframe(xyz.to(DEV), nrm.to(DEV))               # the first frame of a stream: thousands of new boundary voxels
    rows = 64                                                     # a message far too small for them
    left = torch.zeros((1 + rows, 32), dtype=torch.int32, device=DEV)
    right = torch.zeros((1 + rows, 32), dtype=torch.int32, device=DEV)
    note = torch.zeros((8,), dtype=torch.int32).pin_memory()
    m.export_halo_delta(left, right, note)
    torch.cuda.synchronize()
    hl, hr = left[0, :3].cpu().numpy(), right[0, :3].cpu().numpy()
    assert np.array_equal(note.numpy()[:3], hl) and np.array_equal(note.numpy()[4:7], hr)
    assert hl[2] == 1 and hr[2] == 1 and hl[0] <= rows and hr[0] <= rows
    assert max(hl[1], hr[1]) > rows, (hl, hr)
    with pytest.raises(RuntimeError, match="delta halo message"):
        m.n_occupied
    m.n_occupied                                                  # reported once


def _tiled_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from di_fusion_amd.network import utility as net_util
    from di_fusion_amd.stream import FusionStream
    model = net_util.networks_from_arrays(net_util.load_weights_npz())
    scene, cfg = syn.default_room(), syn.MapConfig((-3.2,) * 3, (3.2,) * 3, 0.4)
    st = FusionStream(model, scene, cfg, syn.Intrinsic().scaled(0.25), dev, 4, deg_per_frame=15.0, tiling=(rank, world, None))
    tris = []
    for i in range(4):
        st.step_pipelined(i, d2h="none")
    st.flush("none")
    torch.cuda.synchronize()
    m = st.map
    n = m.n_occupied
    q.put((rank, m._ownership, m.latent_vecs_pos[:n].cpu().numpy(), m.voxel_obs_count[:n].cpu().numpy(), m.latent_vecs[:n].cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL send/recv between ring neighbours)")
def test_spatial_tiling_two_processes_rccl(gpu_model):
    """C5 as it runs in production: one process per GPU, `FusionStream(tiling=...)`, halo exchange over RCCL.  Owned voxels of both
    ranks must equal the single-map stream bit for bit."""
    import socket
    import torch.multiprocessing as mp
    from di_fusion_amd.stream import FusionStream
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tiled_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    cfg = syn.MapConfig((-3.2,) * 3, (3.2,) * 3, 0.4)
    full = FusionStream(gpu_model, syn.default_room(), cfg, syn.Intrinsic().scaled(0.25), DEV, 4, deg_per_frame=15.0)
    for i in range(4):
        full.step(i, d2h="none")
    nF = full.map.n_occupied
    posF = full.map.latent_vecs_pos[:nF].cpu().numpy()
    wF, zF = full.map.voxel_obs_count[:nF].cpu().numpy(), full.map.latent_vecs[:nF].cpu().numpy()
    plane = full.map.n_xyz[1] * full.map.n_xyz[2]
    covered = 0
    for rank, (lo, hi, _), pos, w, z in res:
        own_f, own_s = (posF >= lo * plane) & (posF < hi * plane), (pos >= lo * plane) & (pos < hi * plane)
        of, os_ = np.argsort(posF[own_f]), np.argsort(pos[own_s])
        assert np.array_equal(pos[own_s][os_], posF[own_f][of])
        assert np.array_equal(w[own_s][os_], wF[own_f][of]) and np.array_equal(z[own_s][os_], zF[own_f][of])
        covered += own_f.sum()
    assert covered == nF
