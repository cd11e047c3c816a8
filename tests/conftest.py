import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU case")


def pytest_sessionstart(session):
    """GPU runs: bring torch's HIP runtime up before any test starts child processes (tests/test_multirank_gpu.py) —
    torch failed to find the GPU when it was first initialised after this process had forked children."""
    expr = session.config.getoption("markexpr", "") or ""
    if "not gpu" not in expr:                      # (-m gpu, or a -k selection that may include GPU tests)
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass


@pytest.fixture(scope="session")
def bo():
    """Our C restatement of the reference algorithm (oracle/bbx_oracle.c)."""
    from oracle import ffi
    return ffi.load("bo")


@pytest.fixture(scope="session")
def ref():
    """The compiled reference itself (oracle/_ref); absent on machines without /root/reference
    unless the prebuilt library travelled with the snapshot."""
    from oracle import ffi
    if not ffi.available("ref") and not os.path.isdir("/root/reference"):
        pytest.skip("oracle/_ref not built and no reference tree here")
    return ffi.load("ref")
