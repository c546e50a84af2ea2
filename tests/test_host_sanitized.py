"""The host-only C++ of the library under AddressSanitizer + UndefinedBehaviorSanitizer.

Two pieces of libkicp are plain host C++ shared with test programs: the multi-stream batch driver (kicp_batch.hpp: worker
threads, the pose exchange's rounds, the deadline for peers and the hand-over of workers that never come back) and the
NUMA placement helpers (kicp_numa.hpp: sysfs parsing, mbind / move_pages).  Both test programs (tests/cpp/test_batch_stub.cpp,
test_numa_host.cpp) are rebuilt here with -fsanitize=address,undefined and run: a use after free in the abandoned-worker path,
an overrun in the cpulist parser or the pose blocks aborts with a report.  (Leak checking is off: a batch that was given up
inside an exchange LEAKS its handle by design.)  CPU only."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


@pytest.mark.parametrize("program,args", [("test_batch_stub", []), ("test_numa_host", ["SCRATCH"])])
def test_host_cpp_is_clean_under_asan_and_ubsan(program, args, tmp_path):
    exe = str(tmp_path / (program + "_san"))
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-I" + os.path.join(ROOT, "kiss-icp_amd", "csrc"), "-o", exe, os.path.join(CPP, program + ".cpp"), "-lpthread"]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if b.returncode != 0 and "sanitize" in b.stderr and "unrecognized" in b.stderr:
        pytest.skip("this g++ has no sanitizers")
    assert b.returncode == 0, b.stderr[-3000:]
    scratch = tmp_path / "sysfs"
    scratch.mkdir()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([exe] + [str(scratch) if a == "SCRATCH" else a for a in args], capture_output=True, text=True, env=env, timeout=300, cwd=CPP)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "all checks passed" in r.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
