"""GPU: the MEASURED workload through the MEASURED entry points, frame by frame against the reference.

`tests/golden/seq_c3_long.npz` is the reference stepped over frames 0..24 of the bench stream on BASELINE config C3 — the driver's whole
window (`bench.py --warmup 5 --steps 20`) —, `seq_c3_long_p45.npz` the arc a second stream of a group fuses, `seq_c2_long.npz` a C2 run in which
the 600-count gate (reference map.py:409-410) has frozen 64 % of the voxels by the last frame and cached triangles of re-dirtied voxels are
replaced (map.py:703-714).  The frames are driven exactly as bench.py drives them: frames 0-1 software-pipelined, every later frame by
`FusionStream.step_direct(d2h="dma")` (`dif_integrate_frame` + `dif_extract` with PointSrc, stamps, copy-delivered export), or by
`FusionStreamGroup.step` (`dif_integrate_frames` + `dif_extract_streams`).  Per frame: integer state bit-exact against the reference
(SHA-256 of the arrays), latents <= 2e-5, decoded cubes <= 3e-5 (samples at the refinement threshold excluded and counted, as everywhere),
everything element-wise against the oracle stepped alongside, the frame's triangles against the C marching cubes on the GPU's own cubes, and
the triangles the HOST receives (pinned slot, one frame later) bit-identical to the frame's log rows.  A second pass without any device
synchronisation between frames must hand back the same triangles and leave the same map."""
import numpy as np
import pytest
import torch

from di_fusion_amd import _lib, synthetic as syn
from tests.conftest import GOLDEN
from tests.test_oracle_golden import LONG, check_long_extract_vs_reference, check_long_frame_vs_reference

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
LATENT_TOL = 2e-5
SDF_TOL = 3e-5


def _make(gpu_model, name, n_frames):
    from di_fusion_amd.stream import FusionStream
    make_cfg, phase = LONG[name]
    scene, cfg = make_cfg()
    st = FusionStream(gpu_model, scene, cfg, syn.Intrinsic(), DEV, n_frames, deg_per_frame=0.5, phase_deg=phase)
    return st, scene, cfg, phase


def _drive(st, i, d2h):
    """bench.py's frame_runner: frames 0-1 pipelined (they size the buffers), then direct launches."""
    return st.step_pipelined(i, d2h) if i < 2 else st.step_direct(i, d2h)


def _frame_state(st, f):
    """The map after frame f (device drained by the caller): everything `check_long_frame_vs_reference` wants, as numpy."""
    m = st.map
    c = m._counters.cpu().numpy()
    n = int(c[_lib.C_N_OCCUPIED])
    ok = ~torch.isnan(st.depth[f].view(-1))
    mask = (st.last_unq_mask if f < 2 else st._d_mask).view(-1).to(torch.bool)
    unq = mask[ok].cpu().numpy()
    assert not bool(mask[~ok].any())                         # NaN pixels are never kept
    return c, n, unq, m.indexer.view(-1).cpu().numpy(), m.latent_vecs_pos[:n].cpu().numpy(), m.voxel_obs_count[:n].cpu().numpy(), m.latent_vecs[:n].cpu().numpy()


def _frame_extract(st, c):
    m = st.map
    tens = st.last_tensors if st.last_tensors is not None else m._xbuf[1]
    K, B = int(c[_lib.C_K]), int(c[_lib.C_B])
    lo, hi = int(c[_lib.C_CACHE_KEPT]), int(c[_lib.C_CACHE_T])
    tri, tid, tstd, _ = m._cache
    return (tens["valid_blocks"][:K].cpu().numpy(), tens["occ_slot"][:B].cpu().numpy().astype(np.int64), tens["cube_sdf"][:B].cpu().numpy(),
            tens["cube_std"][:B].cpu().numpy(), (tri[lo:hi].clone(), tid[lo:hi].clone(), tstd[lo:hi].clone()))


def _check_frame(name, g, f, st, om, cfg, O, worst):
    """Frame f of stream `st` (device drained) against the reference fixture and the oracle `om` stepped on the same frame."""
    make_cfg, phase = LONG[name]
    scene, _ = make_cfg()
    c, n, unq, indexer, pos, obs, lat = _frame_state(st, f)
    xyz, nrm = syn.frame_points(scene, f, syn.Intrinsic(), deg_per_frame=0.5, phase_deg=phase)
    omask = om.integrate_keyframe(xyz.numpy(), nrm.numpy())
    # the reference (bit-exact integer state through hashes; the dirty set is checked through valid_blocks below: extract has consumed the flags)
    worst["lat_ref"] = max(worst["lat_ref"], check_long_frame_vs_reference(g, f, unq, n, indexer, pos, obs, None, lat, LATENT_TOL))
    # the oracle, element-wise
    assert np.array_equal(unq, omask) and n == om.n_occupied
    assert np.array_equal(indexer, om.indexer) and np.array_equal(pos, om.latent_vecs_pos[:n]) and np.array_equal(obs, om.voxel_obs_count[:n])
    assert int(c[_lib.C_M]) == om.last_stats["M"] and int(c[_lib.C_C]) == om.last_stats["C"]
    worst["lat_oracle"] = max(worst["lat_oracle"], float(np.abs(lat - om.latent_vecs[:n]).max()))
    assert worst["lat_oracle"] < LATENT_TOL
    vb, occ, cs, cd, new = _frame_extract(st, c)
    oa = om.extract_prepare(4)
    B = occ.shape[0]
    assert np.array_equal(vb, oa["valid_blocks"]) and np.array_equal(occ, oa["occupied_vec_id"])
    flip = np.zeros((B, cs[0].size), dtype=bool)
    if len(oa["near_threshold"]):
        flip[oa["near_threshold"][:, 0], oa["near_threshold"][:, 1]] = True
    worst["cube_ref"] = max(worst["cube_ref"], check_long_extract_vs_reference(g, f, vb, occ, cs, cd, flip, SDF_TOL))
    fl = flip.reshape(cs.shape)
    d = max(float(np.abs(cs - oa["cube_sdf"])[~fl].max()), float(np.abs(cd - oa["cube_std"])[~fl].max()))
    worst["cube_oracle"] = max(worst["cube_oracle"], d)
    worst["flips"] += int(flip.sum())
    assert d < SDF_TOL
    assert abs(int(c[_lib.C_VH]) - oa["n_rows_refine"]) <= int(flip.sum())
    # the frame's triangles: the C marching cubes on the GPU's own cubes
    wt, wi, ws = O.marching_cubes_interp(oa["indexer"], oa["valid_blocks"], oa["vec_batch_mapping"], cs, cd, int(4e6), om.n_xyz, 0.15)
    assert int(c[_lib.C_T]) == wt.shape[0] == new[0].shape[0] > 0
    want = (wt * np.float32(cfg.voxel_size)).astype(np.float32) + om.bound_min
    assert np.array_equal(new[1].cpu().numpy(), wi)
    assert np.abs(new[0].cpu().numpy() - want).max() < 1e-5 and np.abs(new[2].cpu().numpy() - ws).max() < 1e-5
    worst["gated"] = int((obs > 600.0).sum())
    return new


def _new_worst():
    return dict(lat_ref=0.0, lat_oracle=0.0, cube_ref=0.0, cube_oracle=0.0, flips=0, gated=0)


@pytest.mark.parametrize("name,overlap", [("seq_c3_long", True), pytest.param("seq_c3_long", False, marks=pytest.mark.soak),
                                          pytest.param("seq_c2_long", True, marks=pytest.mark.soak)])
def test_direct_dma_stream_vs_reference_frame_by_frame(name, overlap, gpu_model, oracle_net):
    """overlap: the bench's default — frame i+1's integrate front end on a second hardware queue beside frame i's extract
    (`FusionStream.enable_overlap`); False: every frame's launches on one queue.  (`-m gpu` runs the default path on the driver's window;
    the one-queue run and the C2 window with 64 % of the voxels past the 600-count gate are `-m soak`: tools/gpu_soak.sh.)"""
    from oracle import difusion_oracle as O
    g = np.load(GOLDEN / f"{name}.npz")
    F = int(g["n_frames"])
    st, scene, cfg, phase = _make(gpu_model, name, F)
    if overlap and not st.enable_overlap():
        pytest.skip("no second hardware queue to be had in this process (dif_queues_independent)")
    om = O.OracleMap(oracle_net, cfg.bound_min, cfg.bound_max, cfg.voxel_size)
    worst = _new_worst()
    per_frame, delivered = [], []
    for f in range(F):
        out = _drive(st, f, "dma")
        torch.cuda.synchronize()
        if out is not None:
            delivered.append(tuple(x.clone() for x in out))
        per_frame.append(_check_frame(name, g, f, st, om, cfg, O, worst))
    delivered.append(tuple(x.clone() for x in st.flush("dma")))
    n = st.map.n_occupied
    assert np.array_equal(st.map.latent_vecs_pos[:n].cpu().numpy(), g["last_latent_vecs_pos"])
    assert np.array_equal(st.map.voxel_obs_count[:n].cpu().numpy(), g["last_voxel_obs_count"])
    assert np.abs(st.map.latent_vecs[:n].cpu().numpy() - g["last_latent_vecs"]).max() < LATENT_TOL
    print(f"  {name}: {F} frames through step_direct(dma); vs reference: latents {worst['lat_ref']:.1e}, cubes {worst['cube_ref']:.1e}; vs oracle: latents "
          f"{worst['lat_oracle']:.1e}, cubes {worst['cube_oracle']:.1e}; {worst['flips']} samples at the refinement threshold; {worst['gated']} of {n} voxels past the gate")
    assert worst["gated"] > 0.3 * n and st.map.n_deferred == 0
    # what the HOST received (pinned slot, by the copy beside the next frame) is the frame's log rows, bit for bit
    assert len(delivered) == F
    for f, (a, b) in enumerate(zip(per_frame, delivered)):
        assert all(torch.equal(x.cpu(), y.cpu()) for x, y in zip(a, b)), f"frame {f}: delivered triangles differ from the log"
    final = (st.map.indexer.clone(), st.map.latent_vecs[:n].clone(), st.map.voxel_obs_count[:n].clone())
    # second pass: the host never drains the device between frames (the pipeline the bench times)
    del st
    torch.cuda.empty_cache()
    st, _, _, _ = _make(gpu_model, name, F)
    if overlap:
        assert st.enable_overlap()
    got = []
    for f in range(F):
        out = _drive(st, f, "dma")
        if out is not None:
            if f <= 2:
                torch.cuda.synchronize()        # (the two pipelined frames hand back pinned views whose side-stream copy may still be running)
            got.append(tuple(x.clone() for x in out))
    got.append(tuple(x.clone() for x in st.flush("dma")))
    assert len(got) == F
    for f, (a, b) in enumerate(zip(per_frame, got)):
        assert all(torch.equal(x.cpu(), y.cpu()) for x, y in zip(a, b)), f"frame {f}: un-synchronised run differs"
    assert st.map.n_occupied == n
    assert torch.equal(st.map.indexer, final[0]) and torch.equal(st.map.latent_vecs[:n], final[1]) and torch.equal(st.map.voxel_obs_count[:n], final[2])


@pytest.mark.parametrize("n_frames", [8, pytest.param(12, marks=pytest.mark.soak)])
def test_stream_group_two_phases_vs_reference_frame_by_frame(n_frames, gpu_model, oracle_net):
    """S = 2: stream 0 fuses the bench's arc (fixture seq_c3_long), stream 1 the arc starting at 45 degrees (its own fixture): every frame of
    both streams against the reference and the oracle, through `dif_integrate_frames` + `dif_extract_streams`; then the same group again
    without device synchronisation between its frames.  (`-m gpu`: the first 8 frames — the oracle stepped alongside is what takes the time —,
    `-m soak`: all 12 of the second arc's fixture.)"""
    from di_fusion_amd.stream import FusionStreamGroup
    from oracle import difusion_oracle as O
    names = ["seq_c3_long", "seq_c3_long_p45"]
    gs = [np.load(GOLDEN / f"{nm}.npz") for nm in names]
    F = min(n_frames, min(int(g["n_frames"]) for g in gs))

    def build():
        sts = []
        for nm in names:
            st, scene, cfg, phase = _make(gpu_model, nm, F)
            st.map.extract_buffer_bytes = 2 << 30
            sts.append(st)
        return sts, cfg

    streams, cfg = build()
    oms = [O.OracleMap(oracle_net, cfg.bound_min, cfg.bound_max, cfg.voxel_size) for _ in names]
    worst = [_new_worst() for _ in names]
    per_frame = [[] for _ in names]
    delivered = [[] for _ in names]
    grp = None
    for f in range(F):
        if f == 0:
            outs = [st.step(0, "new") for st in streams]           # sizes the buffers; the group takes over from frame 1 (bench.py: GroupBench)
            torch.cuda.synchronize()
            for j, o in enumerate(outs):
                delivered[j].append(tuple(x.clone() for x in o))
            for st in streams:
                st._d_mask_f0 = st.last_unq_mask
        else:
            grp = grp or FusionStreamGroup(streams)
            outs = grp.step(f, "new")
            torch.cuda.synchronize()
            for j, o in enumerate(outs):
                if o is not None:
                    delivered[j].append(tuple(x.clone() for x in o))
        for j, nm in enumerate(names):
            st = streams[j]
            if f == 1:
                st.last_unq_mask = st._d_mask                    # (_frame_state reads `last_unq_mask` for f < 2: the group's frame 1 is a direct frame)
            per_frame[j].append(_check_frame(nm, gs[j], f, st, oms[j], cfg, O, worst[j]))
    for j, o in enumerate(grp.flush("new")):
        delivered[j].append(tuple(x.clone() for x in o))
    for j, nm in enumerate(names):
        print(f"  {nm} in a group of 2: vs reference latents {worst[j]['lat_ref']:.1e}, cubes {worst[j]['cube_ref']:.1e}; vs oracle latents "
              f"{worst[j]['lat_oracle']:.1e}, cubes {worst[j]['cube_oracle']:.1e}")
        # frame 0 was an eager frame (its output is the frame's triangles too); every later frame's delivery is the previous call's
        assert len(delivered[j]) == F
        for f, (a, b) in enumerate(zip(per_frame[j], delivered[j])):
            assert all(torch.equal(x.cpu(), y.cpu()) for x, y in zip(a, b)), f"stream {j} frame {f}: delivered triangles differ from the log"
    finals = [(st.map.n_occupied, st.map.indexer.clone(), st.map.latent_vecs[:st.map.n_occupied].clone()) for st in streams]
    del streams, grp
    torch.cuda.empty_cache()
    streams, _ = build()
    got = [[] for _ in names]
    for j, st in enumerate(streams):
        o = st.step(0, "new")
        torch.cuda.synchronize()
        got[j].append(tuple(x.clone() for x in o))
    grp = FusionStreamGroup(streams)
    for f in range(1, F):
        for j, o in enumerate(grp.step(f, "new")):
            if o is not None:
                got[j].append(tuple(x.clone() for x in o))
    for j, o in enumerate(grp.flush("new")):
        got[j].append(tuple(x.clone() for x in o))
    for j in range(len(names)):
        assert len(got[j]) == F
        for f, (a, b) in enumerate(zip(per_frame[j], got[j])):
            assert all(torch.equal(x.cpu(), y.cpu()) for x, y in zip(a, b)), f"stream {j} frame {f}: un-synchronised group run differs"
        n = streams[j].map.n_occupied
        assert n == finals[j][0] and torch.equal(streams[j].map.indexer, finals[j][1]) and torch.equal(streams[j].map.latent_vecs[:n], finals[j][2])
