"""GPU: the fence-free hand-overs of the product kernels under a litmus load, and the frame export's engine choice and fall-backs.

(1) `dif_test_handoff` (csrc/kernels_litmus.hip.h) hammers the exact pattern of k_sdf_hg_reduce / k_extract_finish — write-through
    stores, `s_waitcnt vmcnt(0)`, then the word; sc1 loads on the other side; no fence — across all XCDs, to pinned host memory, on a stream confined
    to every other CU and beside a kernel that streams through 1 GB: more than 10^6 hand-overs per suite run, none stale, none timed out.
    (tests/test_abi.py checks that the built code object really contains the sc1 / sc0 sc1 instructions the pattern assumes.)
(2) The per-frame export (`d2h="dma"`): SDMA engines the library chooses (only under the HIP runtime that choice was validated with), the runtime's
    own engines, the hipMemcpyAsync fall-back when the HSA runtime cannot be reached, the copy kernel, and a forced re-calibration of the engines —
    every path delivers the same bytes."""
import ctypes

import pytest
import torch

from di_fusion_amd import _lib, synthetic as S

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _litmus(mode, groups, iters, flags):
    out = (ctypes.c_int64 * 4)()
    with torch.cuda.device(DEV):
        rc = _lib.load().dif_test_handoff(mode, groups, iters, flags, out)
    assert rc == 0, rc
    return dict(stale=int(out[0]), handovers=int(out[1]), timeouts=int(out[2]), us=int(out[3]))


def test_device_handovers_are_never_stale():
    total = 0
    for mode, groups, iters in ((0, 256, 1200), (1, 256, 600), (0, 64, 1500), (1, 32, 1500)):
        for flags in (0, 1, 2, 3):
            r = _litmus(mode, groups, iters, flags)
            print(f"  mode {mode} groups {groups} flags {flags}: {r}")
            assert r["stale"] == 0 and r["timeouts"] == 0 and r["handovers"] == groups * iters, (mode, groups, flags, r)
            total += r["handovers"]
    assert total >= 1_000_000


def test_host_handovers_are_never_stale():
    total = 0
    for groups, iters, flags in ((1, 60_000, 0), (8, 15_000, 2), (32, 4_000, 3), (4, 20_000, 1)):
        r = _litmus(2, groups, iters, flags)
        print(f"  {groups} mailboxes x {iters} rounds, flags {flags}: {r}")
        assert r["stale"] == 0 and r["timeouts"] == 0 and r["handovers"] == groups * iters, r
        total += r["handovers"]
    assert total >= 300_000


def test_frame_export_paths_deliver_the_same_bytes(gpu_model):
    from di_fusion_amd.stream import FusionStream
    lib = _lib.load()
    cfg = S.MapConfig((-3.2, -3.2, -3.2), (3.2, 3.2, 3.2), 0.2)
    intr = S.Intrinsic().scaled(0.25)
    F = 8

    def run(sdma_mode=0, force=None):
        prev = lib.dif_test_sdma_mode(sdma_mode)
        try:
            st = FusionStream(gpu_model, S.default_room(), cfg, intr, DEV, F, deg_per_frame=6.0)
            st.force_export = force
            st.step(0, d2h="new")
            torch.cuda.synchronize()
            got = []
            for i in range(1, F):
                o = st.step_direct(i, d2h="dma")
                if o is not None:
                    got.append(tuple(x.clone() for x in o))
            got.append(tuple(x.clone() for x in st.flush("dma")))
            torch.cuda.synchronize()
            return got, st
        finally:
            lib.dif_test_sdma_mode(prev)

    info = (ctypes.c_int32 * 8)()
    want, st = run(force="kernel")                          # the copy kernel (`dif_mesh_cache_export`) is the reference delivery
    assert len(want) == F - 1 and min(w[0].shape[0] for w in want) > 100 and not st.sdma_us
    legs = {"sdma, engines chosen by the library": dict(sdma_mode=0), "sdma, the runtime's engines": dict(sdma_mode=1),
            "runtime unreachable -> hipMemcpyAsync": dict(sdma_mode=2), "every export counted as slow -> engines timed again": dict(sdma_mode=3),
            "hipMemcpyAsync forced": dict(force="blit")}
    for name, kw in legs.items():
        lib.dif_sdma_info(info)
        cal0, exp0 = int(info[3]), int(info[4])
        got, st = run(**kw)
        lib.dif_sdma_info(info)
        assert len(got) == len(want), name
        for f, (a, b) in enumerate(zip(want, got)):
            assert all(torch.equal(x, y) for x, y in zip(a, b)), f"{name}: frame {f + 1} differs"
        print(f"  {name}: ok; runtime {int(info[1])} (validated {int(info[2])}), engines {list(info[5:8])}, calibrations {int(info[3])}, sdma exports {int(info[4])}")
        if kw.get("sdma_mode") == 2:
            assert st.sdma is False and int(info[4]) == exp0           # the call refused, the stream fell back and stayed there
        if kw.get("sdma_mode") in (0, 1, 3) and int(info[0]) == 1:
            assert st.sdma is True and int(info[4]) >= exp0 + F - 1
        if kw.get("sdma_mode") == 1:
            assert list(info[5:8]) == [-1, -1, -1]
        if kw.get("sdma_mode") == 3 and int(info[0]) == 1 and int(info[1]) == int(info[2]):
            assert int(info[3]) > cal0                                  # four "slow" exports in a row: the engines were timed again
