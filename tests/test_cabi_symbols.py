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


def test_staging_narrowing_is_lossless_or_refused():
    """kicp_selftest_narrow = the float64 -> float32 narrowing of the host-input staging path (AVX2 main loop of 8 +
    scalar tail): float32-born values come back bit for bit and are declared exact; a single value that float32
    cannot hold -- at any position of the vector loop or the tail -- makes the whole block inexact (the scan is
    then uploaded as float64); -0.0 and infinities are exact, NaN is not (NaN != NaN: never narrowed)."""
    import ctypes as C

    from kiss_icp_amd import _cabi

    L = _cabi.lib()
    rng = np.random.default_rng(5)

    def narrow(a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        out = np.empty(len(a), dtype=np.float32)
        exact = C.c_int(-1)
        assert L.kicp_selftest_narrow(a.ctypes.data_as(C.c_void_p), len(a), out.ctypes.data_as(C.c_void_p), C.byref(exact)) == 0
        return out, exact.value

    for n in (0, 1, 7, 8, 9, 64, 1003):
        born32 = (rng.normal(0, 50, n).astype(np.float32)).astype(np.float64)
        out, exact = narrow(born32)
        assert exact == 1 and np.array_equal(out.astype(np.float64), born32) and np.array_equal(out, born32.astype(np.float32))
        for pos in range(n):
            if n > 64 and pos % 97:
                continue
            bad = born32.copy()
            bad[pos] = bad[pos] + 1e-9 if bad[pos] != 0 else 1e-50  # more than 24 significant bits / below float32's range
            assert narrow(bad)[1] == 0, (n, pos)
    special = np.array([0.0, -0.0, np.inf, -np.inf, 1.0, 3.4028234663852886e38, 1.401298464324817e-45, 2.0 ** -126])
    out, exact = narrow(special)
    assert exact == 1 and np.array_equal(np.signbit(out), np.signbit(special)) and np.array_equal(out.astype(np.float64), special)
    for v in (np.nan, 1e39, -1e39, 1e-46, 0.1, 16777217.0):
        assert narrow(np.array([1.0, v, 2.0]))[1] == 0, v


def test_every_tuning_knob_is_documented_and_every_documented_knob_exists(cabi):
    """include/kicp.h lists the names kicp_set_option takes; the library's setter is the other side of that list.
    Both directions: a knob added without a word in the header, or documented and gone, fails here.  Each name is also
    SET (to a value its range admits) and an unknown one refused, on a box without a GPU."""
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "kicp.h")).read()
    source = open(os.path.join(root, "kiss-icp_amd", "csrc", "kicp_api.hip")).read()
    documented = set(re.findall(r'"([a-z0-9_]+)"', header[header.index("tuning knobs"):]))
    implemented = set(re.findall(r'strcmp\(name, "([a-z0-9_]+)"\)', source))
    assert documented == implemented, (sorted(documented - implemented), sorted(implemented - documented))
    legal = {"icp_blocks": 0, "icp_points_per_group": 1, "icp_lds_kib": 0, "icp_reserve_cus": 32, "staging_threads": 3,
             "downsample_order": 1, "icp_weight_base": 128, "icp_weight_long_base": 128, "icp_weight_dense_min": 200,
             "icp_weight_dense_div": 1, "icp_weight_quad": -1, "queue_depth": 4, "map_apply_threads": 512,
             "icp_inject_timeout": 0, "icp_inject_timeout_skip": 0, "map_rehash_every": 0, "icp_profile": 0,
             "icp_wide": -1, "icp_wide_prune": 2, "icp_wide_stable": 1, "icp_group_stable": 1, "map_fused_update": 1, "sort_by_rank": 1, "icp_weights_kernel": 1, "icp_wide_prefill": 0, "icp_wide_promote_from": 1,
             "icp_device_streams": 1, "icp_schur_solve": 1, "icp_wide_per_round": 4, "icp_wide_load_eighths": 5, "icp_weight_long_emul": 1, "icp_wide_flat": 3, "icp_wide_group_max": 128, "frame_events": 0,
             "staging_numa_pretend": -1, "wait_timeout_ms": 120000, "inject_stall_ms": 0, "collective_timeout_ms": 1800000}
    L = cabi.lib()
    for name in sorted(implemented):
        assert L.kicp_set_option(name.encode(), legal.get(name, 1)) == 0, name
    assert L.kicp_set_option(b"icp_lds_kib", 64) == 1  # below what the kernel's LDS layout needs
    assert L.kicp_set_option(b"queue_depth", 1) == 1
    assert L.kicp_set_option(b"downsample_order", 2) == 1
