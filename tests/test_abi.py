"""CPU: libdifusion.so loads and exports exactly the symbols include/difusion.h declares (no compute calls)."""
import ctypes
import re

from tests.conftest import ROOT


def header_symbols():
    src = (ROOT / "include" / "difusion.h").read_text()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|int64_t)\s+(dif_\w+)\s*\(", src)))


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
