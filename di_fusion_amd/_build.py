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

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
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
