import os
import sys

import pytest

# (Round 6: every torch REFERENCE operator of the GPU parity tests runs on the CPU -- the round-5 MIOPEN_DEBUG_CONV_IMPLICIT_GEMM=0 work-around
# for a faulting MIOpen solver on the reference side is gone with the MIOpen calls themselves; the product never calls MIOpen.)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
