import os
import sys

import pytest

try:  # torch bundles its own HIP runtime: load it before libmecat_hip.so pulls in /opt/rocm's copy
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (the compiled unmodified reference; this container only)")
