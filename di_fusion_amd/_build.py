"""Build libdifusion.so (HIP, gfx950) in-tree.  Called by `__graft_entry__.build()`; hipcc cross-compiles without a GPU."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libdifusion.so"
SOURCES = [CSRC / "difusion.hip"]
HEADERS = sorted(CSRC.glob("*.hip.h")) + [CSRC / "mc_tables.inc", PKG.parent / "include" / "difusion.h"]

# -fno-slp-vectorize: the SLP vectoriser turns pairs of independent fp32 operations into v_pk_fma_f32 / v_pk_add_f32.  In the bf16-pipe
# MLP kernels (two 250-register waves per SIMD) those packed operations returned wrong values for one 16-lane quarter of a wave now
# and then (about one decoder tile in 10^5; tools/determinism_stress.py reproduces it within seconds), and beside MFMAs they are
# slower than the scalar pair anyway (MI355X_MICROARCH.md, "price of one filler beside MFMAs").
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared",
               "-Wno-unused-result", "-DNDEBUG"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and Path(c).exists():
            return c
    raise RuntimeError("hipcc not found: libdifusion.so cannot be built")


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(p.stat().st_mtime > t for p in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not needs_build():
        return LIB
    cmd = [hipcc()] + HIPCC_FLAGS + [str(s) for s in SOURCES] + ["-o", str(LIB)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force=True)
