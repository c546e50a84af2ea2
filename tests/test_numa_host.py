"""Host placement helpers (kiss-icp_amd/csrc/kicp_numa.hpp): staging memory and helper threads on the GPU's NUMA node.

No reference counterpart -- the reference has no device to be near (replicas, one per GPU; an 8-GPU box has two
sockets).  CPU only: the sysfs look-ups run against a made-up tree, the memory calls against node 0 of this machine (and
may be refused in a container: then the helpers say "unknown" and the library keeps the runtime's own placement)."""
import os
import subprocess


def test_numa_helpers(tmp_path):
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp")
    subprocess.check_call(["make", "-C", d, "test_numa_host"], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(d, "test_numa_host"), str(tmp_path)], capture_output=True, text=True, timeout=120)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout
