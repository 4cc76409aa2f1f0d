import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "zenker-audio-detection_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _torch_device_first():
    """On a GPU box, bring up torch's device context BEFORE libzkast.so has touched the GPU.  torch carries its own HIP
    runtime; initialised late in a process that has already run the library hard (test_model_gpu.py on its own) it can
    report "No HIP GPUs are available", while the other order always works (and is what the full suite happened to do).
    No-op without a GPU: torch.cuda.is_available() is False in the CPU container."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except Exception:
        pass
    yield
