import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mm-diffusion_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """On a GPU box: native backtrace on a fatal signal (libmmd's debugging aid), chained in front of pytest's faulthandler - a host
    SIGSEGV inside the HIP runtime then names the native frame, not only the Python line of the ctypes call."""
    try:
        import torch
        if torch.cuda.is_available():
            from mm_diffusion import _hip
            _hip.lib().mmd_debug_install_crash_handler()
    except Exception as e:          # diagnostics only: never fail the session over it
        print("crash handler not installed:", e)


@pytest.fixture(scope="session")
def gold_dir():
    return GOLD
