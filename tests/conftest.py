import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (the driver runs -m gpu on an MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def dev():
    import torch
    from ipp_amd import capi
    capi.require_gpu()  # fails loudly if the HIP library is missing or no device is visible
    return torch.device("cuda", 0)


@pytest.fixture(scope="session")
def ncc_golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "ncc_golden.npz"))


@pytest.fixture(scope="session")
def psf_golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "psf_golden.npz"))
