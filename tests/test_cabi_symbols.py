"""CPU tests of the drop-in boundary: libkicp.so loads without a GPU, exports every symbol that
include/kicp.h declares, the ctypes table binds exactly that set, and -- with no device -- every
create call fails loudly instead of falling back to a CPU path."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "kicp.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)  # strip comments
    return sorted(set(re.findall(r"\b(kicp_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def cabi():
    from kiss_icp_amd import _cabi

    _cabi.lib()  # ImportError when the extension has not been built: never skipped, never faked
    return _cabi


def test_header_declares_the_expected_surface():
    names = _declared_functions()
    for must in ("kicp_map_create", "kicp_map_closest_neighbor", "kicp_registration_create", "kicp_align_points_to_map",
                 "kicp_pipeline_create", "kicp_pipeline_register_frame", "kicp_voxel_downsample", "kicp_preprocess"):
        assert must in names
    assert len(names) >= 45


def test_library_exports_every_declared_symbol(cabi):
    L = C.CDLL(cabi.LIB_PATH)
    missing = [n for n in _declared_functions() if not hasattr(L, n)]
    assert not missing, missing


def test_ctypes_table_matches_header(cabi):
    bound = set(cabi.SIGNATURES) | set(cabi._STRING_FUNCS)
    assert bound == set(_declared_functions())


def test_exported_symbols_are_plain_c(cabi):
    # the ABI is extern "C": every kicp_* symbol in the dynamic table is unmangled, and nothing of
    # the oracle is linked in
    out = subprocess.run(["nm", "-D", "--defined-only", cabi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    syms = [line.split()[-1] for line in out.splitlines() if line.strip()]
    assert set(_declared_functions()) <= set(syms)
    assert not [s for s in syms if s.startswith("ko_")], "the product must not contain the oracle"


def test_library_does_not_link_the_oracle(cabi):
    out = subprocess.run(["ldd", cabi.LIB_PATH], capture_output=True, text=True).stdout
    assert "kiss_oracle" not in out
    assert "amdhip64" in out  # it is the HIP library, not a CPU stand-in


def test_struct_layouts_match_header(cabi):
    # kicp_config: 4 doubles/ints interleaved as in KISSConfig (pipeline/KissICP.hpp:36-54)
    assert C.sizeof(cabi.Config) == 72
    assert C.sizeof(cabi.IcpStats) == 48
    assert C.sizeof(cabi.FrameStats) == 48 + 48
    c = cabi.Config()
    assert cabi.lib().kicp_config_default(C.byref(c)) == 0
    assert (c.voxel_size, c.max_range, c.min_range, c.max_points_per_voxel) == (1.0, 100.0, 0.0, 20)
    assert (c.min_motion_th, c.initial_threshold, c.max_num_iterations) == (0.1, 2.0, 500)
    assert (c.convergence_criterion, c.max_num_threads, c.deskew) == (0.0001, 0, 1)


def test_version_and_status_strings(cabi):
    mj, mn = C.c_int(-1), C.c_int(-1)
    assert cabi.lib().kicp_version(C.byref(mj), C.byref(mn)) == 0
    assert (mj.value, mn.value) == (0, 3)
    assert cabi.lib().kicp_status_string(0) == b"ok"
    assert b"gfx950" in cabi.lib().kicp_status_string(7)


def test_invalid_arguments_are_statuses_not_crashes(cabi):
    L = cabi.lib()
    assert L.kicp_map_create(1.0, 100.0, 20, 0, None) == 1
    assert L.kicp_config_default(None) == 1
    assert L.kicp_set_option(b"no_such_option", 1) == 1
    assert L.kicp_set_option(None, 1) == 1
    assert L.kicp_pipeline_sync(None) == 1
    assert L.kicp_map_destroy(None) == 0  # destroying nothing is fine, like free(NULL)


def test_no_gpu_means_loud_failure_not_cpu_fallback(cabi):
    """on a box without a GPU every create call returns KICP_ERR_NO_DEVICE"""
    if cabi.device_count() > 0:
        pytest.skip("a GPU is visible here; the no-device path is exercised on CPU-only boxes")
    h = C.c_void_p()
    assert cabi.lib().kicp_map_create(1.0, 100.0, 20, 0, C.byref(h)) == 7 and not h.value
    assert cabi.lib().kicp_registration_create(500, 1e-4, 0, 0, C.byref(h)) == 7 and not h.value
    c = cabi.Config()
    cabi.lib().kicp_config_default(C.byref(c))
    assert cabi.lib().kicp_pipeline_create(C.byref(c), 0, C.byref(h)) == 7 and not h.value
    assert b"no CPU fallback" in cabi.lib().kicp_last_error()
    pts = np.zeros((4, 3))
    n = C.c_size_t(0)
    out = np.empty_like(pts)
    assert cabi.lib().kicp_voxel_downsample(cabi.ptr(pts), 4, 0.5, 0, cabi.ptr(out), C.byref(n)) == 7
    # the Python mirror turns it into an exception
    from kiss_icp_amd.mapping import VoxelHashMap

    with pytest.raises(cabi.KicpError) as e:
        VoxelHashMap(1.0, 100.0, 20)
    assert e.value.status == 7


def test_product_never_imports_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/"""
    pkg = os.path.join(ROOT, "kiss-icp_amd")
    offenders = []
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"kiss_oracle|from oracle|import oracle|ko_[a-z]+_", text):
                    offenders.append(os.path.join(d, f))
    assert not offenders, offenders
