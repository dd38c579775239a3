"""CPU: libdifusion.so loads and exports exactly the symbols include/difusion.h declares (no compute calls)."""
import ctypes
import re

from tests.conftest import ROOT


def header_symbols():
    src = (ROOT / "include" / "difusion.h").read_text()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|int64_t|const char\*)\s+(dif_\w+)\s*\(", src)))


def test_header_library_and_binding_agree():
    from di_fusion_amd import _build, _lib
    lib_path = _build.build()
    syms = header_symbols()
    assert len(syms) >= 15
    h = ctypes.CDLL(str(lib_path))
    for s in syms:
        assert hasattr(h, s), f"{s} declared in difusion.h but not exported"
    assert sorted(_lib.SIGNATURES) == syms
    assert h.dif_version() == 100


def test_library_carries_the_hash_of_the_tree():
    """Build provenance: `dif_build_id()` = hash of csrc/* + include/difusion.h + flags; a library built from other sources is rebuilt,
    whatever the file times say."""
    import os
    from di_fusion_amd import _build
    lib_path = _build.build()
    assert _build.lib_build_id() == _build.source_hash()
    h = ctypes.CDLL(str(lib_path))
    h.dif_build_id.restype = ctypes.c_char_p
    assert h.dif_build_id().decode().split(":")[0] == _build.source_hash()
    assert _build.lib_hipcc_version() == h.dif_build_id().decode().split(":")[1] != ""
    assert not _build.needs_build()
    # a newer file time alone does not trigger a rebuild; a stale id does
    hdr = _build.HEADERS[0]
    st = hdr.stat()
    try:
        os.utime(hdr, (st.st_atime, lib_path.stat().st_mtime + 100))
        assert not _build.needs_build()
    finally:
        os.utime(hdr, (st.st_atime, st.st_mtime))
    assert _build.lib_build_id(_build.PKG / "does_not_exist.so") is None


def test_counter_enum_matches_binding():
    from di_fusion_amd import _lib
    src = (ROOT / "include" / "difusion.h").read_text()
    for name, val in re.findall(r"DIF_C_(\w+)\s*=\s*(\d+)", src):
        if name == "COUNT":
            assert _lib.C_COUNT == int(val)
        else:
            assert getattr(_lib, f"C_{name}") == int(val)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under di_fusion_amd/ may import it."""
    for p in (ROOT / "di_fusion_amd").rglob("*.py"):
        t = p.read_text()
        assert "import oracle" not in t and "from oracle" not in t, p


def test_no_gpu_means_loud_failure():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from di_fusion_amd import synthetic as syn
    from di_fusion_amd.network import utility as net_util
    from di_fusion_amd.system.map import DenseIndexedMap
    model = net_util.networks_from_arrays(net_util.random_weights(0))
    with pytest.raises(RuntimeError):
        DenseIndexedMap(model, syn.config_c1()[1].namespace(), 29, torch.device("cpu"))
    with pytest.raises(RuntimeError):
        model.decoder(torch.zeros((4, 32)))


def _kernel_isa(names):
    """Disassembly of the named kernels (substrings of the mangled names) of the built code object."""
    import subprocess
    import tempfile
    from di_fusion_amd import _build
    llvm = "/opt/rocm/lib/llvm/bin"
    d = _build.build().read_bytes()
    i = d.find(b"\x7fELF", d.find(b"__CLANG_OFFLOAD_BUNDLE__"))
    with tempfile.NamedTemporaryFile(suffix=".o") as f:
        f.write(d[i:]); f.flush()
        syms = subprocess.run([f"{llvm}/llvm-readelf", "-s", "--wide", f.name], capture_output=True, text=True).stdout
        out = {}
        for n in names:
            full = [ln.split()[-1] for ln in syms.splitlines() if n in ln and " FUNC " in ln]
            assert full, f"kernel {n} not in the code object"
            out[n] = subprocess.run([f"{llvm}/llvm-objdump", "-d", "--no-show-raw-insn", f"--disassemble-symbols={full[0]}", f.name],
                                    capture_output=True, text=True).stdout
        return out


def test_no_packed_fp32_arithmetic_in_the_code_object():
    """v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 beside outstanding global loads lose a write now and then on MI355X once the CU's matrix pipes are
    busy (tools/micro/pk_fma_fold.hip reproduces it standalone; csrc/common.hip.h:NO_PACKED_F32): the library is built without the SLP vectoriser and
    the loops the loop vectoriser would pair are marked, so that NO kernel holds one."""
    import subprocess
    import tempfile
    from di_fusion_amd import _build
    d = _build.build().read_bytes()
    i = d.find(b"\x7fELF", d.find(b"__CLANG_OFFLOAD_BUNDLE__"))
    with tempfile.NamedTemporaryFile(suffix=".o") as f:
        f.write(d[i:]); f.flush()
        txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
    assert txt.count("v_mfma_f32_32x32x16_bf16") > 1000                      # (the disassembly is the device code)
    found = re.findall(r"^\s*(v_pk_(?:fma|mul|add)_f32)\b", txt, flags=re.M)
    assert not found, f"{len(found)} packed fp32 instructions in libdifusion.so"
    assert "-fno-slp-vectorize" in _build.HIPCC_FLAGS


def test_fence_free_handovers_compile_to_write_through_stores_and_sc1_loads():
    """The hand-overs inside k_sdf_hg_reduce and k_extract_finish (and their litmus twin, csrc/kernels_litmus.hip.h) rest on what the
    COMPILER makes of relaxed agent- / system-scope atomic stores and loads: write-through (sc1 / sc0 sc1) stores, sc1 loads, no cache write-back or
    invalidate in between.  That is a property of the validated hipcc (di_fusion_amd/_build.py:VALIDATED_HIPCC), checked here in the built code object;
    tests/test_gpu_handoff.py checks on the GPU that the hardware then does what the pattern assumes."""
    isa = _kernel_isa(["k_litmus_device", "k_litmus_host", "k_sdf_hg_reduce", "16k_extract_finishE"])
    dev = isa["k_litmus_device"]
    assert re.search(r"global_store_dwordx2 .* sc1", dev) and re.search(r"global_load_dwordx2 .* sc1", dev) and "s_waitcnt vmcnt(0)" in dev
    host = isa["k_litmus_host"]
    assert re.search(r"global_store_dwordx2 .* sc0 sc1", host) and re.search(r"global_load_dwordx2 .* sc0 sc1", host)
    for k, text in isa.items():
        assert "buffer_wbl2" not in text and "buffer_inv" not in text.replace("s_endpgm", ""), f"{k}: a cache write-back / invalidate inside the kernel body"
    hg = isa["k_sdf_hg_reduce"]
    assert re.search(r"global_store_dwordx2 .* sc1", hg) and re.search(r"global_load_dwordx2 .* sc1", hg) and re.search(r"global_store_dwordx2 .* sc0 sc1", hg)
    assert re.search(r"global_store_dword .* sc0 sc1", isa["16k_extract_finishE"])
