import glob
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "libpll-2_amd"))
sys.path.insert(0, ROOT)

GOLDEN = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")))
GOLDEN = [g for g in GOLDEN if not os.path.basename(g).startswith("model_")]
# derivative fixtures (sumtable + d/dd) are exercised by the *_derivatives tests
DERIV_GOLDEN = [g for g in GOLDEN if "deriv" in os.path.basename(g)]
GOLDEN = [g for g in GOLDEN if g not in DERIV_GOLDEN]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_ids():
    return [os.path.basename(g)[:-4] for g in GOLDEN]


@pytest.fixture(scope="session")
def amd_lib():
    """libpll_amd.so, the product. Fails (not skips) when it was not built."""
    from pllamd import api
    return api.PllLib()


@pytest.fixture(scope="session")
def ref_lib():
    """the real reference (authoring container only)"""
    from pllamd import api
    p = os.path.join(ROOT, "oracle", "_ref", "libpll_ref.so")
    if not os.path.exists(p):
        pytest.skip("oracle/_ref/libpll_ref.so not built (reference sources absent)")
    return api.PllLib(p)
