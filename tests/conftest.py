import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_finish(session):
    """PyTorch-ROCm bundles its own HIP runtime.  If it initialises *after* libporefv_hip.so (linked
    against the system ROCm) has opened the device, it reports "No HIP GPUs are available"; the other
    order works (bench.py imports torch first too).  So: when GPU tests were selected, bring torch's
    runtime up before the first test loads the library."""
    if not any(item.get_closest_marker("gpu") for item in session.items):
        return
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:  # pragma: no cover - torch is only needed by the sharded-driver test
        pass
