import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "soak: long GPU runs of seeds / windows / modes that the default suite covers once — selected only when the "
                                       "marker expression names them (`pytest -m soak`, tools/gpu_soak.sh); never part of `-m gpu` or `-m 'not gpu'`")


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # (`-m gpu` on a machine without a GPU must fail loudly, not silently skip: the product has no CPU fallback.)
    # soak tests run only when asked for by name: the driver's `-m gpu` step has a time limit, and `-m "not gpu"` runs without a GPU
    if "soak" in (config.getoption("-m") or ""):
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("soak") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(scope="session")
def raw_weights():
    import numpy as np
    from di_fusion_amd.network import utility as net_util
    return {k: v for k, v in np.load(net_util.DEFAULT_WEIGHTS).items()}


@pytest.fixture(scope="session")
def oracle_net(raw_weights):
    from oracle import difusion_oracle as O
    return O.OracleNetworks(raw_weights)


@pytest.fixture(scope="session")
def gpu_model(raw_weights):
    """The product default: MLP tiles on the bf16 matrix pipe (six exact slice products per fp32 product, mlp.hip.h "x6")."""
    from di_fusion_amd.network import utility as net_util
    return net_util.networks_from_arrays(raw_weights)


@pytest.fixture(scope="session")
def gpu_model_f32(raw_weights):
    """Same weights with the tiles on the f32-input MFMA (DIF_DECODER_PIPE=f32)."""
    from di_fusion_amd.network import utility as net_util
    return net_util.networks_from_arrays(raw_weights, x6=False)


@pytest.fixture
def mc_grid_cap():
    """Cap the launch of the one-pass marching cubes (ticket mode on small maps) for one test: `mc_grid_cap(n)`; lifted again afterwards."""
    from di_fusion_amd import _lib
    lib = _lib.load()
    yield lambda n: lib.dif_test_mc_grid_cap(int(n))
    lib.dif_test_mc_grid_cap(0)
