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
    from tests import oracle_binding
    oracle_binding.build()
    return oracle_binding


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from ark_analysis_amd import _capi
    _capi.lib()
    return torch.device("cuda:0")


@pytest.fixture(params=[pytest.param("oracle"), pytest.param("hip", marks=pytest.mark.gpu)])
def som_backend(request, monkeypatch):
    """Runs a host-logic test twice: on CPU with the device entry points of
    ark_analysis_amd.flowsom swapped for the oracle (test infrastructure; `-m "not gpu"`), and on
    the GPU box with the real HIP path (`-m gpu`)."""
    if request.param == "hip":
        import torch
        if not torch.cuda.is_available():
            pytest.skip("no HIP device")
        return "hip"
    from tests import oracle_backend
    oracle_backend.install(monkeypatch.setattr)
    return "oracle"
