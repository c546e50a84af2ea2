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
    config.addinivalue_line("markers", "cold_libs: first use loads a library of hundreds of MB (RCCL: 570 MB of compressed code objects read in "
                            "full; torch) -- run LAST")


def pytest_collection_modifyitems(config, items):
    """GPU boxes boot from an image that is read on demand, and how fast differs by two orders of magnitude between leases
    (profiles/r05_x_first_rccl_probe.txt: librccl.so's 570 MB at 5 MB/s = 115 s; 400 MB/s on another lease).  The tests that
    pull such a library in run after everything else, so that a slow lease costs them minutes, not the parity tests their
    place in the run (the driver stops at the first failure or at its own time limit)."""
    early = [it for it in items if it.get_closest_marker("cold_libs") is None]
    late = [it for it in items if it.get_closest_marker("cold_libs") is not None]
    items[:] = early + late


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
