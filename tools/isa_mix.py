#!/usr/bin/env python3
"""Instruction mix of a kernel of libdifusion.so from its ISA (MFMA : other VALU : LDS : VMEM : SMEM : SALU : waitcnt / barriers).
   python tools/isa_mix.py k_encodeILb1E [k_decode_voxelsILb1E ...]      (substring of the mangled name; first match)"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
LLVM = Path("/opt/rocm/lib/llvm/bin")


def disassemble(so):
    d = Path(so).read_bytes()
    i = d.find(b"\x7fELF", d.find(b"__CLANG_OFFLOAD_BUNDLE__"))
    with tempfile.NamedTemporaryFile(suffix=".o") as f:
        f.write(d[i:]); f.flush()
        return subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_cvt_pk_bf16"): return "valu_slice_cvt"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    return "other"


def main():
    txt = disassemble(ROOT / "di_fusion_amd" / "libdifusion.so")
    for want in sys.argv[1:]:
        m = re.search(r"^[0-9a-f]+ <([^>]*%s[^>]*)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)" % re.escape(want), txt, re.S | re.M)
        if not m:
            print(want, ": not found"); continue
        ops = [l.split()[0] for l in m.group(2).splitlines() if l.startswith("\t")]
        mix = {}
        for o in ops:
            mix[classify(o)] = mix.get(classify(o), 0) + 1
        n_mfma = max(1, mix.get("mfma", 0))
        print(f"{m.group(1)[:60]}: {len(ops)} instructions; " + ", ".join(f"{k} {v}" for k, v in sorted(mix.items(), key=lambda kv: -kv[1]))
              + f"; non-MFMA per MFMA {(len(ops) - n_mfma) / n_mfma:.2f}")


if __name__ == "__main__":
    main()
