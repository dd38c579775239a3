"""Bounded, seeded runs of the differential fuzzers (tools/fuzz_*.py) as tests: random small problems, product (HIP, through the C ABI)
against the oracle.  Bars as in the tools: bit-exact integer state / indices / masks, latents <= 2e-5, SDF <= 5e-5, vertices <= 1e-5.
The fuzzers found the one real product bug of round 1 (a workgroup-level LDS table that filled up on scattered points); a regression
there now fails the suite."""
import sys

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu
sys.path.insert(0, str(ROOT / "tools"))


# `-m gpu`: one seed per fuzzer (30 / 50 / 30 / 5 cases); `-m soak` (tools/gpu_soak.sh) the other seeds
SEEDS = [0] + [pytest.param(s, marks=pytest.mark.soak) for s in (1, 2, 3)]


@pytest.mark.parametrize("seed", SEEDS)
def test_fuzz_integrate_extract_query(seed):
    import fuzz_integrate
    fuzz_integrate.run(cases=30, seed=seed)


@pytest.mark.parametrize("seed", SEEDS)
def test_fuzz_marching_cubes(seed):
    import fuzz_mc
    fuzz_mc.run(cases=50, seed=seed)


@pytest.mark.parametrize("seed", SEEDS)
def test_fuzz_cloud_ops(seed):
    import fuzz_cloud
    fuzz_cloud.run(cases=30, seed=seed)


@pytest.mark.parametrize("seed", SEEDS[:3])
def test_fuzz_stream_group(seed):
    """S streams through one chain of launches against the same streams alone: bit-identical maps, meshes and per-frame updates on random
    configurations (tools/fuzz_group.py)."""
    import fuzz_group
    fuzz_group.run(cases=5, seed=seed)
