import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# The parity tests compare the HIP path with the oracle bit for bit, which is the contract of the EXACT arithmetic flavour of the device
# code (luminary_amd/csrc/device/flavour.h). The library's own default is the fast flavour; tests/test_flavours.py gates that one against
# exact and selects flavours through the API.
os.environ["LUM_FLAVOUR"] = "exact"
# A host tiles whole-frame renders over every visible GPU (luminary_amd/csrc/host/api.cpp); the single-device tests pin it to one so that
# they mean the same on any box. tests/test_multi_gpu.py lifts the cap where it tests the tiled path.
os.environ["LUM_MAX_DEVICES"] = "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:
        return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly rather than pass on a fallback; nothing is skipped here on purpose.
    return
