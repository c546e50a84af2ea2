"""CPU tests of bench.py's launch logic: `--gpus N` must mean N streams -- as one rank of N under torch.distributed.run
(what the driver launches for N > 1), or, started plainly, N streams driven in-process through the C-ABI's batch entry --
and must fail loudly, never fall back to a single stream, when the box cannot provide them."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_one_gpu_is_a_single_stream():
    assert bench.launch_plan(1, None, -1, 1) == ("single", [0], None)
    assert bench.launch_plan(1, None, 3, 8) == ("single", [3], None)


def test_under_a_launcher_this_process_is_one_rank():
    assert bench.launch_plan(8, "8", -1, 8)[0] == "rank"
    assert bench.launch_plan(1, "1", -1, 1)[0] == "rank"
    with pytest.raises(SystemExit, match="disagree"):
        bench.launch_plan(2, "4", -1, 8)


def test_without_a_launcher_n_streams_run_in_process():
    assert bench.launch_plan(8, None, -1, 8) == ("in-process", list(range(8)), "rccl")
    assert bench.launch_plan(2, None, -1, 8) == ("in-process", [0, 1], "rccl")
    # streams stacked on one device (1-GPU box): RCCL refuses two ranks per device, so the exchange is the host communicator
    assert bench.launch_plan(2, None, 0, 1) == ("in-process", [0, 0], "host")


def test_too_few_devices_is_an_error_not_a_single_stream_run():
    with pytest.raises(SystemExit, match="needs 4 GPUs, 1 visible"):
        bench.launch_plan(4, None, -1, 1)
    with pytest.raises(SystemExit):
        bench.launch_plan(0, None, -1, 1)


def test_bench_refuses_n_gpus_on_a_box_without_them():
    """the real command on this (GPU-less) box: it stops before generating a single scan"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=120)
    from kiss_icp_amd import _cabi

    if _cabi.device_count() >= 8:
        pytest.skip("this box has 8 GPUs")
    assert r.returncode != 0
    assert "needs 8 GPUs" in r.stderr
    assert '"n_gpus"' not in r.stdout


def test_bench_has_no_undefined_module_level_names():
    """every global name a function of bench.py loads is defined in the module (a helper deleted by an edit shows up here,
    not in the one GPU call that would have used it)"""
    import ast
    import builtins

    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    defined = set(dir(builtins))
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)):
            defined.add(node.name)
        elif isinstance(node, (ast.Import, ast.ImportFrom)):
            defined.update((a.asname or a.name).split(".")[0] for a in node.names)
        elif isinstance(node, ast.Assign):
            defined.update(t.id for t in node.targets if isinstance(t, ast.Name))
        elif isinstance(node, ast.For):
            defined.update(n.id for n in ast.walk(node.target) if isinstance(n, ast.Name))
    for fn in [n for n in tree.body if isinstance(n, ast.FunctionDef)]:
        local = {a.arg for a in ast.walk(fn) if isinstance(a, ast.arg)}
        for n in ast.walk(fn):
            if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store):
                local.add(n.id)
            elif isinstance(n, (ast.Import, ast.ImportFrom)):
                local.update((a.asname or a.name).split(".")[0] for a in n.names)
            elif isinstance(n, (ast.FunctionDef, ast.Lambda)) and n is not fn:
                if isinstance(n, ast.FunctionDef):
                    local.add(n.name)
            elif isinstance(n, ast.ExceptHandler) and n.name:
                local.add(n.name)
        missing = sorted({n.id for n in ast.walk(fn) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load)} - local - defined)
        assert not missing, (fn.name, missing)


class _Done:
    def __init__(self, returncode, stdout):
        self.returncode, self.stdout = returncode, stdout.encode()


def test_bench_measures_in_its_own_process_once():
    """a bench whose process dies has failed: no supervising parent, no second attempt (round 4 had both)"""
    assert not hasattr(bench, "supervise")
    src = open(bench.__file__).read()
    assert "KICP_BENCH_CHILD" not in src and "attempts" not in src


def test_cpu_baseline_has_a_march_native_leg_that_never_fails_the_bench():
    """the port built with -march=native on the box that runs it, timed in a child process on the same
    sample: a figure beside the generic build's -- or an error string, never an exception"""
    sys.path.insert(0, os.path.join(ROOT, "kiss-icp_amd", "python"))
    from kiss_icp_amd.datasets import generate_scans

    factory, ds_kw, cfg_over, _ = bench.workload("kitti")
    scans = generate_scans(factory, dict(ds_kw, n_frames=3), range(3))
    r = bench.cpu_baseline_native(scans, 1, 2, cfg_over, 4)
    assert "error" not in r, r
    assert r["value"] > 0 and "-march=native" in r["flags"] and "-ffp-contract=off" in r["flags"] and r["cores"] == 4
    bad = bench.cpu_baseline_native(scans, 1, 2, dict(cfg_over, no_such_option=1), 4)
    assert "error" in bad and "value" not in bad
