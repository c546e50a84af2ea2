"""The algebra header shared by the HIP kernels and libkicp's host side (kicp_math.hpp), checked on
the host: divide-free voxel coordinates vs the reference's floor(p / voxel_size), voxel key packing,
SE(3) exp/log/inverse consistency, the pivoted 6x6 LDLT incl. Eigen's zero-pivot rule.  The test
program is compiled with hipcc but calls no HIP API, so it runs without a GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_math_header_on_the_host():
    d = os.path.join(ROOT, "tests", "cpp")
    subprocess.check_call(["make", "-C", d, "test_math_host"], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(d, "test_math_host")], capture_output=True, text=True, timeout=120)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout and "0 mismatches" in r.stdout
