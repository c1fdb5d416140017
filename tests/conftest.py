import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    # tests/test_emu_parity.py re-runs a selection of the GPU tier's small cases, unchanged, against the host functional
    # model of the library (tools/emu): same C ABI, same binding, same assertions
    if os.environ.get("KS_TESTS_ON_FUNCTIONAL_MODEL") == "1" and os.environ.get("KS_HIP_LIB", "").endswith("libks_hip_emu.so"):
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
