import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Tests of this process keep reads in torch tensors and run torch.distributed next to the library: the process says so
# before libkasa_hip.so is loaded, so that there is ONE HIP runtime in it (kasa_amd/capi.py:share_torch_runtime).  Hosts
# without torch -- the C++ driver, smoke(), tools/fuzz_gpu.py -- run on ROCm's own runtime.
try:
    from kasa_amd import capi as _capi
    _capi.share_torch_runtime()
except Exception:                                                # pragma: no cover
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
