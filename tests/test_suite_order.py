"""The GPU suite's order is part of its robustness: the tests whose first use pulls in a library of hundreds of MB (RCCL: one
570 MB compressed bundle read in full; torch) are marked `cold_libs` and run LAST (tests/conftest.py), because a lease whose
image arrives at 5 MB/s makes them wait minutes (profiles/r05_x_first_rccl_probe.txt) and the driver stops at the first
failure or at its own time limit.  Checked here on the CPU by collecting the GPU suite."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _collected(extra):
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/", "-m", "gpu", "--collect-only", "-q"] + extra, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return [line.strip() for line in r.stdout.splitlines() if "::" in line]


def test_tests_that_load_rccl_or_torch_run_last():
    every = _collected([])
    cold = set(_collected(["-m", "gpu and cold_libs"]))
    assert len(every) > 100 and 3 <= len(cold) <= 12, (len(every), sorted(cold))
    tail = every[len(every) - len(cold):]
    assert set(tail) == cold, (tail, sorted(cold))
    # everything that creates a batch WITHOUT a host communicator (i.e. loads RCCL) or imports torch is among them
    for path in ("tests/test_gpu_paths.py", "tests/test_gpu_deadlines.py", "tests/test_cpp_api.py", "tests/test_gpu_parity.py"):
        src = open(os.path.join(ROOT, path)).read()
        for m in re.finditer(r"^def (test_\w+)\(.*?(?=^def |^@pytest|\Z)", src, re.S | re.M):
            name, body = m.group(1), m.group(0)
            loads = ("import torch" in body or "dlpack_device_check" in body or "--batch-only" in body or
                     (re.search(r"StreamBatch\(", body) and "comm=" not in body))
            if loads:
                assert any(t.split("::")[1].split("[")[0] == name for t in cold), "%s::%s loads RCCL / torch but is not marked cold_libs" % (path, name)
    assert every[0].endswith("test_cpp_program_against_oracle")  # (the C++ program's front half stays the suite's first test)
