import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rgbid-slam_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One rgbid context for the GPU tests; fails loudly (no fallback) when the HIP library or device is missing."""
    import torch
    from rgbid import device
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    c = device.Context(0)
    yield c
    c.close()
