import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _gpu_tests_run_without_autograd(request):
    """The HIP path is forward-only (its outputs are tied to a backward that raises, runtime.forward_only): GPU tests
    are inference, so they run with autograd off, like the reference's examples (torch.inference_mode)."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import torch
    with torch.no_grad():
        yield
