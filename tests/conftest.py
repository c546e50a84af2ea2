import os
import sys

import pytest

# the oracle's "all cores" default is counter-productive on many-core hosts (see oracle/kiss_oracle.c)
os.environ.setdefault("KISS_ORACLE_THREADS", "8")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kiss-icp_amd", "python")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def _gpu_count():
    try:
        from kiss_icp_amd import _cabi

        return _cabi.device_count()
    except Exception:
        return 0


@pytest.fixture(scope="session")
def gpu():
    """skip-proof guard: -m gpu tests must run on a box with a GPU and the built library"""
    from kiss_icp_amd import _cabi

    if not os.path.exists(_cabi.LIB_PATH):  # e.g. a fresh checkout on the GPU box: build it (hipcc is in the image)
        import subprocess

        subprocess.check_call(["make", "-C", os.path.join(ROOT, "kiss-icp_amd", "csrc")])
    _cabi.lib()  # raises ImportError when libkicp.so is missing -- never fall back
    n = _cabi.device_count()
    if n == 0:
        pytest.fail("gpu-marked test running without a visible GPU")
    return n
