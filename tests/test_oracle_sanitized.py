"""The oracle under AddressSanitizer + UndefinedBehaviorSanitizer ("sanitizer targets").

`make -C oracle sanitize` builds the same C restatement with -fsanitize=address,undefined; its own CPU tests
(tests/test_oracle.py: the restatement against naive_ref, the goldens, scipy) then run on THAT library in a child process --
LD_PRELOAD of the sanitizer runtime (the interpreter is not instrumented), KISS_ORACLE_LIB selecting the build.  Any
out-of-bounds access, use after free or undefined operation of the checker aborts the child with a report."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_asan_and_ubsan():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan for this gcc")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "sanitize"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, LD_PRELOAD=asan, KISS_ORACLE_LIB=os.path.join(ROOT, "oracle", "_san", "libkiss_oracle.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               KISS_ORACLE_THREADS="4")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, env=env, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
