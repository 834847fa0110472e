import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs the read-only reference checkout")


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir(REFERENCE)
    skip_ref = pytest.mark.skip(reason="/root/reference not present on this machine")
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    skip_gpu = pytest.mark.skip(reason="no HIP device on this machine (run with -m gpu on the MI355X box)")
    for item in items:
        if "reference" in item.keywords and not have_ref:
            item.add_marker(skip_ref)
        if "gpu" in item.keywords and not have_gpu:
            item.add_marker(skip_gpu)


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """Every test session starts from a freshly built (or up-to-date) C-ABI library."""
    from cream_amd import build
    if not build.is_current():
        try:
            build.build()
        except Exception:
            # On the GPU box the prebuilt .so travels with the snapshot; if hipcc is
            # unavailable there we still want the tests to run against it.
            if not os.path.exists(build.LIB):
                raise
    from oracle import rpe_index_oracle
    rpe_index_oracle.build()
    yield
