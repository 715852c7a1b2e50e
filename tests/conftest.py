import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): oracle/spec_oracle.py over spec_oracle.c."""
    from oracle import spec_oracle
    spec_oracle.build()
    return spec_oracle


@pytest.fixture(scope="session")
def svc():
    """One SpectralService (GPU context) for the whole session; fails loudly without the HIP library."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test started without a GPU"
    from spectral_analyzer_amd import SpectralService
    s = SpectralService(0)
    yield s
    s.close()
