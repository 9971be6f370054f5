import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "nano-vllm-hw3_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "quant_golden.npz"))


@pytest.fixture(scope="session")
def triton_golden():
    """Outputs of the reference's @triton.jit kernels under Triton's CPU interpreter
    (tests/golden/make_golden_triton.py)."""
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "triton_golden.npz"))
