"""GPU: the ways `FusionStream` drives a frame (eager, software-pipelined, direct launches through the device-readable frame descriptor on
one and on two hardware queues, stream groups) must leave bit-identical maps and hand back the same triangles."""
import numpy as np
import pytest
import torch

from di_fusion_amd import synthetic as S

S_ = S          # (tests below use `S` for the number of streams)

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
N_FRAMES = 6


def make_stream(gpu_model, initial_capacity=1 << 13):
    from di_fusion_amd.stream import FusionStream
    cfg = S.MapConfig((-3.2, -3.2, -3.2), (3.2, 3.2, 3.2), 0.2)          # 32^3 grid
    intr = S.Intrinsic().scaled(0.25)
    return FusionStream(gpu_model, S.default_room(), cfg, intr, DEV, N_FRAMES, deg_per_frame=6.0, initial_capacity=initial_capacity)


def snapshot(st):
    m = st.map
    n = m.n_occupied
    tri, tid, tstd = m.mesh_cache_tensors(new_only=False)
    return dict(n=n, indexer=m.indexer.clone(), latent=m.latent_vecs[:n].clone(), obs=m.voxel_obs_count[:n].clone(),
                tri=tri.clone(), tid=tid.clone(), tstd=tstd.clone())


def same(a, b):
    assert a["n"] == b["n"] > 100
    for k in ("indexer", "latent", "obs", "tri", "tid", "tstd"):
        assert torch.equal(a[k], b[k]), k
    assert a["tri"].shape[0] > 1000


def test_eager_pipelined_and_direct_agree(gpu_model):
    outs = {}
    st = make_stream(gpu_model)
    per_frame = []
    for i in range(N_FRAMES):
        o = st.step(i, d2h="new")
        per_frame.append(tuple(x.clone() for x in o))
    torch.cuda.synchronize()
    outs["eager"] = snapshot(st)

    st = make_stream(gpu_model)
    got = []
    for i in range(N_FRAMES):
        o = st.step_pipelined(i, d2h="new")
        if o is not None:
            torch.cuda.synchronize()
            got.append(tuple(x.clone() for x in o))
    o = st.flush()
    got.append(tuple(x.clone() for x in o))
    outs["pipelined"] = snapshot(st)
    assert len(got) == N_FRAMES
    for a, b in zip(per_frame, got):
        assert all(torch.equal(x, y) for x, y in zip(a, b))

    same(outs["eager"], outs["pipelined"])

    # direct launches through the frame descriptor (two C calls per frame), with a forced compaction
    st = make_stream(gpu_model)
    got = []
    st.step(0, d2h="new")
    torch.cuda.synchronize()
    got.append(per_frame[0])
    for i in range(1, N_FRAMES):
        if i == 4:
            st.map._gc_wanted = True
        o = st.step_direct(i, d2h="new")
        if o is not None:
            torch.cuda.synchronize()
            got.append(tuple(x.clone() for x in o))
    o = st.flush()
    got.append(tuple(x.clone() for x in o))
    assert len(got) == N_FRAMES
    for a, b in zip(per_frame[1:], got[1:]):
        assert all(torch.equal(x, y) for x, y in zip(a, b))
    same(outs["eager"], snapshot(st))


@pytest.mark.parametrize("d2h", ["dma", "none"])
def test_two_queue_frames_are_bit_identical(d2h, gpu_model):
    """`enable_overlap`: frame i+1's integrate front end (unproject ... encoder) on a second hardware queue beside frame i's extract, its
    fusion kernel behind that extract, every extract behind its frame's fusion kernel (device-side waits on words the kernels publish:
    dif_map_t.frame_seq).  Every frame's triangles and the final map equal the eager single-queue run bit for bit — with new voxels
    allocated on every frame (the slots frame i+1 allocates must stay invisible to frame i's extract), across a forced mesh-log compaction,
    a capacity growth, an eager frame in the middle (leaves and re-enters the mode), and with the host never draining the device."""
    st = make_stream(gpu_model)
    F = N_FRAMES
    per_frame = []
    for i in range(F):
        o = st.step(i, d2h="new")
        torch.cuda.synchronize()
        per_frame.append(tuple(x.clone() for x in o))
    ref = snapshot(st)
    for rep in range(3):
        st = make_stream(gpu_model, initial_capacity=(1 << 13) if rep < 2 else None)
        if not st.enable_overlap():
            pytest.skip("no second hardware queue to be had in this process (dif_queues_independent)")
        got = []
        st.step(0, d2h="new")
        torch.cuda.synchronize()
        got.append(per_frame[0])
        for i in range(1, F):
            if i == 3 and rep == 0:
                st.map._gc_wanted = True                        # a mesh-log compaction with an overlapped frame in flight
            if i == 4 and rep == 1:                             # an eager frame between overlapped ones
                o = st.step_pipelined(i, d2h="new")
            else:
                o = st.step_direct(i, d2h=d2h)
            if o is not None:
                if d2h == "none" or (rep == 1 and i in (4, 5)):
                    torch.cuda.synchronize()                    # ("none": device views of the log; around the eager frame: side-stream copies)
                got.append(tuple(x.clone() for x in o))
        o = st.flush(d2h)
        torch.cuda.synchronize()
        got.append(tuple(x.clone() for x in o))
        assert len(got) == F
        for f, (a, b) in enumerate(zip(per_frame[1:], got[1:])):
            assert all(torch.equal(x.cpu(), y.cpu()) for x, y in zip(a, b)), f"rep {rep} frame {f + 1}"
        same(ref, snapshot(st))
        assert st.queues_independent is True
        if rep == 0:
            assert st.map._gc_epoch == 1


def test_deferred_export_rides_with_the_next_frame(gpu_model):
    """`step_direct` with the stream's own capacity (room for the frames in flight, so the host never completes a frame early): a frame's new
    triangles are copied to the pinned slot by the leading workgroups of the NEXT frame's point kernels (`dif_map_t.pending_export`); the
    last frame's by `dif_export_pending` at flush.  Every frame's
    triangles and the final map equal the eager stream's, bit for bit; no frame was exported by the fallback path in mid-stream."""
    st = make_stream(gpu_model)
    per_frame = []
    for i in range(N_FRAMES):
        o = st.step(i, d2h="new")
        torch.cuda.synchronize()
        per_frame.append(tuple(x.clone() for x in o))
    want = snapshot(st)
    st = make_stream(gpu_model, initial_capacity=None)
    assert st.map._capacity >= 3 * 7 * (19200 // 17)
    got, early = [], 0
    orig = st._export_deferred_now

    def counting(handle):
        nonlocal early
        if isinstance(handle, dict) and handle.get("deferred") and "export_event" not in handle and not handle.get("export_carried"):
            early += 1
        return orig(handle)
    st._export_deferred_now = counting
    st.step(0, d2h="new")
    torch.cuda.synchronize()
    got.append(per_frame[0])
    for i in range(1, N_FRAMES):
        o = st.step_direct(i, d2h="new")
        if o is not None:
            torch.cuda.synchronize()
            got.append(tuple(x.clone() for x in o))
    in_stream = early
    o = st.flush()
    got.append(tuple(x.clone() for x in o))
    assert len(got) == N_FRAMES
    for f, (a, b) in enumerate(zip(per_frame[1:], got[1:])):
        assert all(torch.equal(x, y) for x, y in zip(a, b)), f"frame {f + 1}"
    same(want, snapshot(st))
    assert in_stream == 0 and early <= 1                   # only the last frame (at flush) needed the stand-alone copy


@pytest.mark.parametrize("grouped", [False, True], ids=["one_stream", "group_of_three"])
def test_copy_engine_delivery_equals_kernel_delivery(grouped, gpu_model):
    """d2h="dma": a direct frame's new triangles go to its pinned slot by the copy engine (`dif_mesh_cache_export_dma` on a side stream, once
    the frame's stamp has been seen) instead of riding with the next frame's kernels — the same triangles, frame by frame, and the same map,
    also across a forced compaction of the mesh log and mixed with kernel-delivered frames."""
    from di_fusion_amd.stream import FusionStreamGroup
    S = 3 if grouped else 1

    def run(mode_of):
        streams = [make_stream(gpu_model, initial_capacity=None) for _ in range(S)]
        got = [[] for _ in range(S)]
        for j, st in enumerate(streams):
            got[j].append(_eager(st, 0))
        grp = FusionStreamGroup(streams) if grouped else None
        for i in range(1, N_FRAMES):
            if i == 3:
                for st in streams:
                    st.map._gc_wanted = True               # a compaction of the log between two frames
            outs = grp.step(i, d2h=mode_of(i)) if grouped else [streams[0].step_direct(i, d2h=mode_of(i))]
            torch.cuda.synchronize()        # (a frame finished under another mode than it was launched with is delivered by the async fallback copy)
            for j, o in enumerate(outs):
                if o is not None:
                    got[j].append(tuple(x.clone() for x in o))
        outs = grp.flush(mode_of(N_FRAMES)) if grouped else [streams[0].flush(mode_of(N_FRAMES))]
        for j, o in enumerate(outs):
            got[j].append(tuple(x.clone() for x in o))
        return got, [snapshot(st) for st in streams]

    want, want_map = run(lambda i: "new")
    for mode_of in (lambda i: "dma", lambda i: "dma" if i % 2 else "new"):
        got, got_map = run(mode_of)
        for j in range(S):
            assert len(got[j]) == len(want[j]) == N_FRAMES
            for f, (a, b) in enumerate(zip(want[j], got[j])):
                assert a[0].shape[0] > 0 and all(torch.equal(x, y) for x, y in zip(a, b)), f"stream {j} frame {f}"
            same(want_map[j], got_map[j])


@pytest.mark.parametrize("other", ["pipelined", "eager"])
def test_compaction_with_a_deferred_export_pending(other, gpu_model):
    """A `step_direct` frame leaves its triangle export pending (absolute log rows); the NEXT frame is driven by another stepping mode on
    a frame where the mesh log is compacted.  The pending copy must be carried out before the compaction moves the rows (ADVICE r3:
    `step_pipelined` / `step` compacted first and the direct frame's triangles came out wrong)."""
    ref = make_stream(gpu_model)
    want = [_eager(ref, i) for i in range(N_FRAMES)]
    want_state = snapshot(ref)
    st = make_stream(gpu_model, initial_capacity=None)
    got = {0: _eager(st, 0)}
    pending = None                                          # index of the frame whose output the next pipelined call hands back

    def take(o, idx):
        if o is not None:
            torch.cuda.synchronize()
            got[idx] = tuple(x.clone() for x in o)

    for i in range(1, N_FRAMES):
        if i % 2 == 1:
            take(st.step_direct(i, d2h="new"), pending)
            pending = i
        else:
            st.map._gc_wanted = True                        # the frame behind a direct frame compacts the log
            if other == "pipelined":
                take(st.step_pipelined(i, d2h="new"), pending)
                pending = i
            else:
                take(st.step(i, d2h="new"), i)              # eager: its own output; the direct frame stays pending
    take(st.flush(), pending)
    assert st.map._gc_epoch >= 2
    assert sorted(got) == list(range(N_FRAMES))
    for f in range(N_FRAMES):
        assert all(torch.equal(x, y) for x, y in zip(want[f], got[f])), f"frame {f}"
    same(want_state, snapshot(st))


def test_host_staging_overflow_falls_back(gpu_model):
    """A frame with more new triangles than its pinned staging slot holds must still hand back all of them (through the side-stream export)."""
    ref = make_stream(gpu_model)
    want = [tuple(x.clone() for x in ref.step(i, d2h="new")) for i in range(4)]
    st = make_stream(gpu_model)
    st.HOST_OUT_TRIANGLES = 64                               # instance attribute shadows the class default before the slots are made
    st.step(0, d2h="new")
    got = [want[0]]
    for i in range(1, 4):
        o = st.step_direct(i, d2h="new")
        if o is not None:
            torch.cuda.synchronize()
            got.append(tuple(x.clone() for x in o))
    o = st.flush()
    torch.cuda.synchronize()
    got.append(tuple(x.clone() for x in o))
    assert len(got) == 4 and min(g[0].shape[0] for g in got) > 64
    for a, b in zip(want, got):
        assert all(torch.equal(x, y) for x, y in zip(a, b))


def test_frame_descriptor_entry_point_bit_exact(gpu_model):
    import ctypes
    import struct
    from di_fusion_amd import _lib
    intr = S.Intrinsic().scaled(0.25)
    R, t = S.orbit_pose(3, deg_per_frame=6.0)
    depth, ncam = S.render_frame(S.default_room(), R, t, intr, DEV)
    H, W = intr.height, intr.width
    Rc = (ctypes.c_float * 9)(*[float(np.float32(v)) for v in R.reshape(-1)])
    tc = (ctypes.c_float * 3)(*[float(np.float32(v)) for v in t])
    a_xyz, a_n = torch.empty((H * W, 3), device=DEV), torch.empty((H * W, 3), device=DEV)
    b_xyz, b_n = torch.empty((H * W, 3), device=DEV), torch.empty((H * W, 3), device=DEV)
    lib = _lib.load()
    _lib.check(lib.dif_unproject_transform(_lib.ptr(depth), _lib.ptr(ncam), _lib.ptr(a_xyz), _lib.ptr(a_n), H, W, intr.fx, intr.fy, intr.cx, intr.cy,
                                           Rc, tc, _lib.stream_ptr()), "dif_unproject_transform")
    desc = torch.frombuffer(bytearray(struct.pack("<QQ12f", depth.data_ptr(), ncam.data_ptr(), *Rc, *tc)), dtype=torch.uint8).to(DEV)
    _lib.check(lib.dif_unproject_transform_frame(_lib.ptr(desc), _lib.ptr(b_xyz), _lib.ptr(b_n), H, W, intr.fx, intr.fy, intr.cx, intr.cy,
                                                 _lib.stream_ptr()), "dif_unproject_transform_frame")
    nan = lambda x: torch.nan_to_num(x, nan=123.0)
    assert torch.equal(nan(a_xyz), nan(b_xyz)) and torch.equal(nan(a_n), nan(b_n))
    assert lib.dif_unproject_transform_frame(None, _lib.ptr(b_xyz), _lib.ptr(b_n), H, W, 1.0, 1.0, 0.0, 0.0, _lib.stream_ptr()) != 0


def test_c3_full_size_invariants(gpu_model):
    """BASELINE config C3 at full size (128^3 grid, 640x480 frames): size-independent properties (the frame-by-frame comparison of the same
    stream with the reference and the oracle is tests/test_gpu_long.py): slot <-> voxel bijection, conservation of the observation count, idempotence of extract, vertices inside their voxel,
    and bit-identity between an eager run and a directly launched run (which walks the frame in 16x16 pixel tiles) — with per-voxel extract
    buffers far smaller than the map (`DenseIndexedMap._extract_rows`) and no extract deferred."""
    from di_fusion_amd.stream import FusionStream
    scene, cfg = S.config_c3()
    finals = []
    for rep in range(2):
        st = FusionStream(gpu_model, scene, cfg, S.Intrinsic(), DEV, 5, deg_per_frame=0.5)
        rows = 0
        for i in range(5):
            if rep == 1 and i >= 1:                     # second run: direct launches through dif_integrate_frame (16x16 pixel tiles)
                st.step_direct(i, d2h="none")
            else:
                st.step(i, d2h="none")
        st.flush("none")
        rows = sum(s["M"] for s in st.stats)
        m = st.map
        n = m.n_occupied
        assert n > 15000
        assert m.n_deferred == 0 and m._xbuf[0][1] <= (1 << 16) < m._capacity      # <= 0.5 GB of per-voxel extract buffers for the 524,288-slot map
        pos = m.latent_vecs_pos[:n]
        idx = m.indexer.view(-1)
        assert torch.equal(idx[pos], torch.arange(n, device=DEV))                     # indexer[pos[s]] == s
        assert int((idx >= 0).sum()) == n and int(idx.max()) == n - 1                  # and nothing else is allocated
        assert float(m.voxel_obs_count[:n].double().sum()) == float(rows)              # every gathered row is counted exactly once
        tri, tid, tstd = m.mesh_cache_tensors()
        assert tri.shape[0] > 100000
        vox = torch.stack([tid // (128 * 128), (tid // 128) % 128, tid % 128], -1).float()
        lo = torch.tensor(cfg.bound_min, device=DEV) + vox * cfg.voxel_size
        eps = 1e-4
        assert bool(((tri >= (lo - eps)[:, None, :]) & (tri <= (lo + cfg.voxel_size + eps)[:, None, :])).all())
        assert bool((tstd <= 0.15).all())                                               # max_std gate (mc_interp_kernel.cu:304)
        assert bool((idx[tid] >= 0).all())                                              # triangles belong to allocated voxels
        # nothing is dirty any more: another extract is a no-op
        before = tri.clone()
        m.extract_mesh_arrays(4, int(4e6), max_std=0.15, to_host=False)
        assert m.last_counters["K"] == 0 and m.last_counters["T"] == 0
        assert torch.equal(m.mesh_cache_tensors()[0], before)
        finals.append((m.latent_vecs[:n].clone(), before, tid.clone()))
    assert all(torch.equal(a, b) for a, b in zip(finals[0], finals[1]))


def _eager(st, i):
    """One eager frame; its new triangles (an async copy into pinned memory) cloned once the copy is complete."""
    o = st.step(i, d2h="new")
    torch.cuda.synchronize()
    return tuple(x.clone() for x in o)


def _solo_and_group(gpu_model, make, S, n_frames, before_group=None):
    """Every stream alone (eager: the reference run of this file), then the same S streams as one group; returns per-stream
    (per-frame outputs, final snapshot) of both."""
    from di_fusion_amd.stream import FusionStreamGroup
    solo = []
    for j in range(S):
        st = make(j)
        per = [_eager(st, i) for i in range(n_frames)]
        solo.append((per, snapshot(st)))
        del st
    torch.cuda.empty_cache()
    if before_group is not None:
        before_group()
    streams = [make(j) for j in range(S)]
    got = [[_eager(st, 0)] for st in streams]      # sizes the buffers; the group takes over from frame 1
    grp = FusionStreamGroup(streams)
    for i in range(1, n_frames):
        outs = grp.step(i, d2h="new")
        torch.cuda.synchronize()
        for j, o in enumerate(outs):
            if o is not None:
                got[j].append(tuple(x.clone() for x in o))
    for j, o in enumerate(grp.flush()):
        got[j].append(tuple(x.clone() for x in o))
    return solo, [(got[j], snapshot(streams[j])) for j in range(S)], streams


@pytest.mark.parametrize("S", [1, 3, 4])
def test_stream_group_matches_single_streams(S, gpu_model):
    """S independent subsequences (different arcs of the orbit, private maps) whose frames share their twelve launches
    (`dif_integrate_frames` + `dif_extract_streams`): every stream's per-frame triangles and final map equal that stream stepped alone,
    bit for bit — including across a forced mesh-log compaction of one stream and a capacity growth that one stream pulls the others
    through."""
    from di_fusion_amd.stream import FusionStream
    cfg = S_.MapConfig((-3.2, -3.2, -3.2), (3.2, 3.2, 3.2), 0.2)
    intr = S_.Intrinsic().scaled(0.25)

    def make(j):
        return FusionStream(gpu_model, S_.default_room(), cfg, intr, DEV, N_FRAMES, deg_per_frame=6.0, phase_deg=45.0 * j, initial_capacity=None)

    solo, grp, streams = _solo_and_group(gpu_model, make, S, N_FRAMES)
    for j in range(S):
        assert len(grp[j][0]) == N_FRAMES
        for f, (a, b) in enumerate(zip(solo[j][0], grp[j][0])):
            assert all(torch.equal(x, y) for x, y in zip(a, b)), f"stream {j} frame {f}"
        same(solo[j][1], grp[j][1])
    if S > 1:       # the streams really are different subsequences
        assert not torch.equal(solo[0][1]["indexer"], solo[1][1]["indexer"])


def test_stream_group_marching_cubes_in_ticket_mode(gpu_model, mc_grid_cap):
    """The grouped launch of the one-pass marching cubes capped at five workgroups per stream (dif_test_mc_grid_cap): every stream's groups of four
    voxels are claimed through its ticket counter — the path of a map with thousands of dirty voxels — and the triangles still equal the
    streams stepped alone with a workgroup per group, bit for bit."""
    from di_fusion_amd.stream import FusionStream
    cfg = S_.MapConfig((-3.2, -3.2, -3.2), (3.2, 3.2, 3.2), 0.2)
    intr = S_.Intrinsic().scaled(0.25)

    def make(j):
        return FusionStream(gpu_model, S_.default_room(), cfg, intr, DEV, N_FRAMES, deg_per_frame=6.0, phase_deg=45.0 * j, initial_capacity=None)

    solo, grp, streams = _solo_and_group(gpu_model, make, 3, N_FRAMES, before_group=lambda: mc_grid_cap(5))
    print("  dirty voxels of the last frame:", [st.map.last_counters["K"] for st in streams], "(ticket mode above 20)")
    for j in range(3):
        assert len(grp[j][0]) == N_FRAMES
        for f, (a, b) in enumerate(zip(solo[j][0], grp[j][0])):
            assert all(torch.equal(x, y) for x, y in zip(a, b)), f"stream {j} frame {f}"
        same(solo[j][1], grp[j][1])


def test_stream_group_survives_compaction_and_growth(gpu_model):
    from di_fusion_amd.stream import FusionStream, FusionStreamGroup
    cfg = S_.MapConfig((-3.2, -3.2, -3.2), (3.2, 3.2, 3.2), 0.2)
    intr = S_.Intrinsic().scaled(0.25)

    def make(j):
        return FusionStream(gpu_model, S_.default_room(), cfg, intr, DEV, N_FRAMES, deg_per_frame=6.0, phase_deg=45.0 * j, initial_capacity=None)

    solo = []
    for j in range(2):
        st = make(j)
        per = [_eager(st, i) for i in range(N_FRAMES)]
        solo.append((per, snapshot(st)))
    streams = [make(j) for j in range(2)]
    got = [[_eager(st, 0)] for st in streams]
    grp = FusionStreamGroup(streams)
    for i in range(1, N_FRAMES):
        if i == 2:
            streams[1].map._gc_wanted = True                    # one stream compacts its mesh log with a deferred export pending
        if i == 4:
            with streams[0].map._state_lock:                    # one stream's map grows: the group re-shapes every map
                streams[0]._export_deferred_now(streams[0]._pending)
                streams[0].map._alloc_state(2 * streams[0].map._capacity)
        outs = grp.step(i, d2h="new")
        torch.cuda.synchronize()
        for j, o in enumerate(outs):
            if o is not None:
                got[j].append(tuple(x.clone() for x in o))
    for j, o in enumerate(grp.flush()):
        got[j].append(tuple(x.clone() for x in o))
    assert streams[1].map._gc_epoch == 1 and streams[0].map._capacity == streams[1].map._capacity
    for j in range(2):
        assert len(got[j]) == N_FRAMES
        for f, (a, b) in enumerate(zip(solo[j][0], got[j])):
            assert all(torch.equal(x, y) for x, y in zip(a, b)), f"stream {j} frame {f}"
        same(solo[j][1], snapshot(streams[j]))


def test_stream_group_growth_after_every_slot_was_used_without_device_sync(gpu_model):
    """a map that grows pulls the others of its group through a re-allocation BETWEEN `_direct_begin` and the launches — their slot
    descriptors are rebuilt and must get the frame's stamp, notify word and output fields again (`FusionStream._direct_fill`).  With a stale
    stamp of 0 the host would read a slot's previous counters and triangles at once.  Here the growth comes at frame 6, when all four slots
    have been used (their stamp words hold old, non-zero stamps), and the host never drains the device between the group's frames."""
    from di_fusion_amd.stream import FusionStream, FusionStreamGroup
    cfg = S_.MapConfig((-3.2, -3.2, -3.2), (3.2, 3.2, 3.2), 0.2)
    intr = S_.Intrinsic().scaled(0.25)
    F = 9

    def make(j):
        return FusionStream(gpu_model, S_.default_room(), cfg, intr, DEV, F, deg_per_frame=6.0, phase_deg=45.0 * j, initial_capacity=None)

    solo = []
    for j in range(2):
        st = make(j)
        per = [_eager(st, i) for i in range(F)]
        solo.append((per, snapshot(st)))
    streams = [make(j) for j in range(2)]
    got = [[_eager(st, 0)] for st in streams]
    grp = FusionStreamGroup(streams)
    for i in range(1, F):
        if i == 6:
            with streams[0].map._state_lock:                    # stream 0 grows on its own; the group's step pulls stream 1 along
                streams[0]._export_deferred_now(streams[0]._pending)
                streams[0].map._alloc_state(2 * streams[0].map._capacity)
        outs = grp.step(i, d2h="new")
        for j, o in enumerate(outs):                            # (no torch.cuda.synchronize(): what comes back must be complete by itself)
            if o is not None:
                got[j].append(tuple(x.clone() for x in o))
    for j, o in enumerate(grp.flush()):
        got[j].append(tuple(x.clone() for x in o))
    assert streams[0].map._capacity == streams[1].map._capacity == 2 * make(0).map._capacity
    for j in range(2):
        assert len(got[j]) == F
        for f, (a, b) in enumerate(zip(solo[j][0], got[j])):
            assert all(torch.equal(x, y) for x, y in zip(a, b)), f"stream {j} frame {f}"
        same(solo[j][1], snapshot(streams[j]))


def test_stream_group_rejects_maps_it_cannot_batch(gpu_model):
    """`dif_extract_streams` needs capacity > 4096 and one extract-buffer size for all maps; the group says so instead of failing
    with an opaque EINVAL at the first step."""
    from di_fusion_amd.stream import FusionStream, FusionStreamGroup
    cfg = S_.MapConfig((-3.2, -3.2, -3.2), (3.2, 3.2, 3.2), 0.2)
    intr = S_.Intrinsic().scaled(0.25)
    small = [FusionStream(gpu_model, S_.default_room(), cfg, intr, DEV, 2, deg_per_frame=6.0, phase_deg=45.0 * j, initial_capacity=2048) for j in range(2)]
    for st in small:
        st.step(0, "new")
    torch.cuda.synchronize()
    grp = FusionStreamGroup(small)
    grp.step(1, "new")                                          # the group grows such maps to the minimum instead of failing
    grp.flush()
    assert all(st.map._capacity >= 8192 for st in small)
    a = FusionStream(gpu_model, S_.default_room(), cfg, intr, DEV, 2, deg_per_frame=6.0)
    b = FusionStream(gpu_model, S_.default_room(), cfg, intr, DEV, 2, deg_per_frame=6.0, phase_deg=45.0)
    b.map.extract_buffer_bytes = 1 << 26                        # -> fewer rows per extract buffer than stream a
    for st in (a, b):
        st.step(0, "new")
    torch.cuda.synchronize()
    with pytest.raises(ValueError, match="extract_buffer_bytes"):
        FusionStreamGroup([a, b]).step(1, "new")


def test_stream_group_c3_four_streams(gpu_model):
    """BASELINE config C3 (128^3 grid, 640x480 frames), four subsequences starting 45 degrees apart, 6 frames each, as one group against
    each stream alone: per-frame mesh updates and final maps bit-identical."""
    from di_fusion_amd.stream import FusionStream
    scene, cfg = S_.config_c3()

    def make(j):
        st = FusionStream(gpu_model, scene, cfg, S_.Intrinsic(), DEV, 6, deg_per_frame=0.5, phase_deg=45.0 * j)
        st.map.extract_buffer_bytes = 1 << 30                   # (eight maps live in this test: keep the per-voxel extract buffers at 1 GB each)
        return st

    solo, grp, streams = _solo_and_group(gpu_model, make, 4, 6)
    for j in range(4):
        assert len(grp[j][0]) == 6
        for f, (a, b) in enumerate(zip(solo[j][0], grp[j][0])):
            assert all(torch.equal(x, y) for x, y in zip(a, b)), f"stream {j} frame {f}"
        a, b = solo[j][1], grp[j][1]
        assert a["n"] == b["n"] > 10000
        for k in ("indexer", "latent", "obs", "tri", "tid", "tstd"):
            assert torch.equal(a[k], b[k]), (j, k)


def test_direct_frames_with_and_without_point_arrays(gpu_model):
    """`dif_integrate_frame` with xyz_world / normal_world (the first kernel writes every pixel's world point and normal, the later stages
    read them) and without (the later stages recompute the few points they need from the depth pixel): identical maps and meshes, and the
    arrays — when asked for — equal the stand-alone unproject + transform of the same frame."""
    import ctypes
    from di_fusion_amd import _lib
    snaps = {}
    for keep in (False, True):
        st = make_stream(gpu_model, initial_capacity=None)
        st.keep_points = keep
        if keep:
            st.xyz.fill_(7.0); st.nrm.fill_(7.0)
        st.step(0, d2h="new")
        for i in range(1, N_FRAMES):
            st.step_direct(i, d2h="new")
        st.flush()
        snaps[keep] = snapshot(st)
        if keep:
            intr = st.intr
            R, t = st.poses[N_FRAMES - 1]
            want_xyz, want_nrm = torch.empty_like(st.xyz), torch.empty_like(st.nrm)
            _lib.check(_lib.load().dif_unproject_transform(_lib.ptr(st.depth[N_FRAMES - 1]), _lib.ptr(st.ncam[N_FRAMES - 1]), _lib.ptr(want_xyz), _lib.ptr(want_nrm),
                                                           intr.height, intr.width, intr.fx, intr.fy, intr.cx, intr.cy, R, t, _lib.stream_ptr()), "dif_unproject_transform")
            nan = lambda x: torch.nan_to_num(x, nan=123.0)
            assert torch.equal(nan(st.xyz), nan(want_xyz)) and torch.equal(nan(st.nrm), nan(want_nrm))
    same(snaps[False], snaps[True])


def test_stream_group_argument_checks_and_mesh_left_in_hbm(gpu_model):
    """The batched entry points refuse what they cannot run (a map twice in one batch, maps of different capacity, a spatially tiled map),
    and a group with the mesh left in HBM (d2h = "none": no export, no notify) ends in the same maps as the streams alone."""
    import ctypes
    from di_fusion_amd import _lib
    from di_fusion_amd.stream import FusionStream, FusionStreamGroup
    cfg = S_.MapConfig((-3.2, -3.2, -3.2), (3.2, 3.2, 3.2), 0.2)
    intr = S_.Intrinsic().scaled(0.25)
    mk = lambda j, **kw: FusionStream(gpu_model, S_.default_room(), cfg, intr, DEV, 4, deg_per_frame=6.0, phase_deg=45.0 * j, **kw)
    solo = []
    for j in range(2):
        st = mk(j)
        for i in range(4):
            st.step(i, d2h="none")
        torch.cuda.synchronize()
        solo.append(snapshot(st))
    streams = [mk(j) for j in range(2)]
    for st in streams:
        st.step(0, d2h="none")
    grp = FusionStreamGroup(streams)
    for i in range(1, 4):
        assert grp.step(i, d2h="none") is not None
    grp.flush("none")
    for j in range(2):
        same(solo[j], snapshot(streams[j]))
    # --- refusals (DIF_EINVAL = -1), straight at the C ABI ---
    lib = _lib.load()
    a, b = streams
    w = a.map.model.packed.weights_struct(DEV)
    frames = (_lib.DifStreamFrame * 2)()
    H, W = intr.height, intr.width

    def fill(f, st, cmap=None):
        sl = st._d_slots[0]
        f.map = ctypes.pointer(cmap if cmap is not None else st.map._cmap)
        f.frame_dev = _lib.ptr(sl["frame"])
        f.xyz_world, f.normal_world = _lib.ptr(None), _lib.ptr(None)
        f.unq_mask = _lib.ptr(st._d_mask)
        f.ws, f.ws_bytes = _lib.ptr(st.map._ws), st.map._ws.numel()
        f.buf = ctypes.pointer(st._d_bufs[0])
    with torch.cuda.device(DEV):
        sp = _lib.stream_ptr()
        fill(frames[0], a); fill(frames[1], a)                                  # the same map twice
        assert lib.dif_integrate_frames(frames, 2, ctypes.byref(w), H, W, intr.fx, intr.fy, intr.cx, intr.cy, sp) == -1
        assert lib.dif_integrate_frames(frames, 9, ctypes.byref(w), H, W, intr.fx, intr.fy, intr.cx, intr.cy, sp) == -1      # more than DIF_MAX_STREAMS
        odd = _lib.DifMap.from_buffer_copy(b.map._cmap)
        odd.capacity = b.map._cmap.capacity // 2                                # maps of different capacity
        fill(frames[1], b, odd)
        assert lib.dif_integrate_frames(frames, 2, ctypes.byref(w), H, W, intr.fx, intr.fy, intr.cx, intr.cy, sp) == -1
        tiled = _lib.DifMap.from_buffer_copy(b.map._cmap)
        tiled.own_x_lo, tiled.own_x_hi, tiled.halo = 0, b.map.n_xyz[0] // 2, 3   # a spatially tiled map
        fill(frames[1], b, tiled)
        assert lib.dif_integrate_frames(frames, 2, ctypes.byref(w), H, W, intr.fx, intr.fy, intr.cx, intr.cy, sp) == -1
        assert lib.dif_extract_streams(frames, 2, ctypes.byref(w), 5, 0.15, 1, sp) == -1                                      # resolution > 4
    with pytest.raises(ValueError):
        FusionStreamGroup([])
    torch.cuda.synchronize()


@pytest.mark.parametrize("overlap", [False, True], ids=["one_queue", "two_queues"])
def test_extract_defers_instead_of_overflowing_and_catches_up(overlap, gpu_model):
    """The per-voxel extract buffers of a stream are sized by what its frames decode, not by the map's capacity.  An extract whose dirty set
    could need more rows than there are (min(7 K, n_occupied) > rows) changes NOTHING on the device — K = B = T = 0, the dirty set is kept,
    counters[DIF_C_DEFERRED] says how many rows it wanted — the host grows the buffers when it sees that frame's counters, and the next
    extract meshes everything that accumulated.  That is the reference's own semantics for a caller that integrates several frames between two
    `extract_mesh` calls (map.py:303-308: the dirty set accumulates): map and mesh cache are bit-identical to an eager run that integrates
    frames 1 and 2 WITHOUT extracting.  The synchronous façade call (`extract_mesh_arrays`) grows and retries inside the call."""
    ref = make_stream(gpu_model)
    for i in range(N_FRAMES):
        if i in (1, 2):                                             # integrate only: what the deferred frames amount to
            ref._unproject(i)
            ref.map.integrate_keyframe(ref.xyz, ref.nrm)
        else:
            ref.step(i, d2h="none")
    torch.cuda.synchronize()
    want = snapshot(ref)

    st = make_stream(gpu_model, initial_capacity=None)
    if overlap and not st.enable_overlap():
        pytest.skip("no second hardware queue to be had in this process (dif_queues_independent)")
    st.step(0, d2h="none")
    real = st.map._extract_rows
    st.map._extract_rows = lambda res, no_cache=False: 256          # far too few rows for frames 1 and 2
    empty = []
    for i in range(1, N_FRAMES):
        if i == 3:
            st.map._extract_rows = real                             # the host has seen a deferral by now: the real sizing grows the buffers
        o = st.step_direct(i, d2h="none")
        if o is not None:
            empty.append(int(o[0].shape[0]))
    st.flush("none")
    torch.cuda.synchronize()
    assert st.map.n_deferred == 2 and empty[0] == 0 and empty[1] == 0 and min(empty[2:]) > 0
    assert st.map._xbuf[0][1] >= 2 * st.map._extract_rows_wanted > 512
    same(want, snapshot(st))
    assert st.map.last_counters["deferred"] == 0 and int(st.map._dirty[:want["n"]].sum()) == 0

    # the synchronous call: starts with 256 rows, finds them too few, grows, retries — one call, the whole mesh, every frame
    full = make_stream(gpu_model)
    for i in range(N_FRAMES):
        full.step(i, d2h="none")
    torch.cuda.synchronize()
    st = make_stream(gpu_model, initial_capacity=None)
    st.map.MIN_EXTRACT_ROWS = 256
    for i in range(N_FRAMES):
        st.step(i, d2h="none")
    torch.cuda.synchronize()
    assert st.map.n_deferred >= 1
    same(snapshot(full), snapshot(st))
    # a limit that no growth can satisfy is an error (with the dirty set kept), not an endless deferral
    st = make_stream(gpu_model, initial_capacity=None)
    st.map.MIN_EXTRACT_ROWS = st.map.EXTRACT_ROWS_FLOOR = 64
    st.map.extract_buffer_bytes = 1 << 20                           # 128 rows
    with pytest.raises(RuntimeError, match="extract_buffer_bytes"):
        for i in range(N_FRAMES):
            st.step(i, d2h="none")
    torch.cuda.synchronize()
    st.map.extract_buffer_bytes = 8 << 30
    st.map.extract_mesh_arrays(4, int(4e6), max_std=0.15, to_host=False)           # nothing was lost
    assert st.map.last_counters["K"] > 100 and st.map.last_counters["deferred"] == 0
