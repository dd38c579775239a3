"""Build libdifusion.so (HIP, gfx950) in-tree.  Called by `__graft_entry__.build()`; hipcc cross-compiles without a GPU."""
from __future__ import annotations

import hashlib
import os
import re
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libdifusion.so"
SOURCES = [CSRC / "difusion.hip"]
HEADERS = sorted(CSRC.glob("*.hip.h")) + [CSRC / "mc_tables.inc", PKG.parent / "include" / "difusion.h"]

# -fno-slp-vectorize: the SLP vectoriser turns pairs of independent fp32 operations into v_pk_fma_f32 / v_pk_add_f32.  A v_pk_fma_f32 that runs
# while other global loads of its wave are still outstanding loses, now and then, its write to the low register of the destination pair in lanes
# 48..63 once other waves keep the CU's matrix pipes busy — the lattice kernel's fold constants, 5-13 voxels of 12,765 per launch with the round-6
# kernels (tools/determinism_stress.py under DIF_LIB=<a build without this flag>: every repeat differs).  Reproduced outside this library by
# tools/micro/pk_fma_fold.hip (never with v_fma_f32, never behind s_waitcnt vmcnt(0), never without MFMA waves on the CU;
# profiles/r06_experiments.md 5).  The loop vectoriser makes the same instructions: csrc/common.hip.h:NO_PACKED_F32 marks those loops, and
# tests/test_abi.py checks that the code object holds no packed fp32 arithmetic at all.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared",
               "-Wno-unused-result", "-DNDEBUG"]


# The compiler this tree was validated with (parity suite + tools/determinism_stress.py soak).  Two things in the kernels are properties of THIS
# compiler rather than of the language: (1) the schedule of the bf16-pipe MLP tiles (below), and (2) the fence-free hand-overs — relaxed
# agent-/system-scope atomic stores and loads must come out as write-through (sc1) stores and sc1 loads, with `s_waitcnt vmcnt(0)` in between
# (csrc/kernels_litmus.hip.h; k_sdf_hg_reduce, k_extract_finish): tests/test_gpu_handoff.py hammers that pattern on every
# `pytest -m gpu`, and tests/test_abi.py checks the instructions in the built code object.  The bf16-pipe MLP kernels once showed a
# schedule-dependent wrong tile with the SLP vectoriser on (see above): a different compiler means a different schedule, so its version is
# part of the build id, and a build with another one says so loudly (DIF_ALLOW_OTHER_HIPCC=1 silences the warning once the GPU suite and
# the soak have passed with it).
VALIDATED_HIPCC = "7.2.26015"


def hipcc_version() -> str:
    try:
        out = subprocess.run([hipcc(), "--version"], capture_output=True, text=True, timeout=60).stdout
        m = re.search(r"HIP version:\s*([0-9.]+)", out)
        return m.group(1) if m else "unknown"
    except Exception:
        return "unknown"


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and Path(c).exists():
            return c
    raise RuntimeError("hipcc not found: libdifusion.so cannot be built")


def source_hash() -> str:
    """Hash of everything the library is compiled from (sources, headers, tables, the C ABI header, the compiler flags)."""
    h = hashlib.sha256()
    for p in sorted(SOURCES + HEADERS, key=lambda q: q.name):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()[:16]


def lib_build_id(lib: Path = None):
    """The hash the library carries (`dif_build_id()`), read from the file without loading it; None if absent."""
    lib = LIB if lib is None else lib
    if not lib.exists():
        return None
    m = re.search(rb"DIF_BUILD_ID=([0-9a-f]{16})", lib.read_bytes())
    return m.group(1).decode() if m else None


def lib_hipcc_version(lib: Path = None):
    """The compiler version the library says it was built with (`dif_build_id()` = "<source hash>:<hipcc version>")."""
    lib = LIB if lib is None else lib
    if not lib.exists():
        return None
    m = re.search(rb"DIF_BUILD_ID=[0-9a-f]{16}:([0-9A-Za-z.\-]+)", lib.read_bytes())
    return m.group(1).decode() if m else None


def needs_build() -> bool:
    """The shipped library is used only if it was built from exactly this tree (file times say nothing after a checkout or rsync)."""
    return lib_build_id() != source_hash()


def build(force: bool = False, verbose: bool = True, shared: bool = False) -> Path:
    """Compile into a temporary file and rename it onto LIB under an exclusive file lock: several processes that find a stale library at
    once (torchrun ranks, multi-process tests) must neither compile onto a file another one is dlopen-ing nor each pay for a build
    (`shared`: a forced build is satisfied by one that another process finished while this one waited)."""
    if not force and not needs_build():
        return LIB
    import fcntl
    with open(str(LIB) + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not needs_build() and (not force or shared):
                return LIB                       # another process built it while this one waited for the lock
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose: bool) -> Path:
    ver = hipcc_version()
    if not ver.startswith(VALIDATED_HIPCC) and os.environ.get("DIF_ALLOW_OTHER_HIPCC") != "1":
        print(f"WARNING: libdifusion is being built with hipcc {ver}; the tree was validated with {VALIDATED_HIPCC}. Run `pytest -m gpu` and "
              "`python tools/determinism_stress.py 400` on the GPU before trusting the MLP kernels of this build.", flush=True)
    tmp = LIB.with_name(f".{LIB.name}.{os.getpid()}.tmp")
    cmd = [hipcc()] + HIPCC_FLAGS + [f'-DDIF_BUILD_ID="{source_hash()}:{ver}"'] + [str(s) for s in SOURCES] + ["-o", str(tmp)]
    if verbose:
        print(" ".join(cmd), flush=True)
    try:
        subprocess.check_call(cmd)
        if lib_build_id(tmp) != source_hash():
            raise RuntimeError("libdifusion.so was built but does not carry the hash of this tree")
        os.replace(tmp, LIB)                     # atomic: a concurrent dlopen sees the old file or the new one, never a partial write
    finally:
        if tmp.exists():
            tmp.unlink()
    return LIB


if __name__ == "__main__":
    build(force=True)
