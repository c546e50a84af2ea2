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


def test_a_bench_child_that_dies_is_started_once_more(monkeypatch, capsys):
    """`python bench.py` measures in a child process: a device fault aborts the process that caused it, and a dead bench
    prints nothing.  One more attempt, the line says which one it was; a refusal (status 3) or a usage error (2) is not
    repeated; a second death is the bench's failure."""
    import json

    calls = []

    def fake_run(outcomes):
        it = iter(outcomes)

        def run(cmd, env=None, stdout=None):
            calls.append((cmd, env.get("KICP_BENCH_CHILD")))
            return next(it)
        return run

    line = json.dumps({"metric": "RegisterFrame scans/s", "value": 1.0})
    monkeypatch.setattr(subprocess, "run", fake_run([_Done(-6, "noise\n"), _Done(0, "[Gloo] chatter\n" + line + "\n")]))
    assert bench.supervise(["--steps", "2"]) == 0
    out, err = capsys.readouterr()
    assert json.loads(out.strip().splitlines()[-1]) == {"metric": "RegisterFrame scans/s", "value": 1.0, "attempts": 2}
    assert out.splitlines()[0] == "[Gloo] chatter" and "status -6" in err and "once more" in err
    assert len(calls) == 2 and all(c[1] == "1" and c[0][-2:] == ["--steps", "2"] for c in calls)

    calls.clear()
    monkeypatch.setattr(subprocess, "run", fake_run([_Done(0, line + "\n")]))
    assert bench.supervise([]) == 0
    assert json.loads(capsys.readouterr().out.strip())["attempts"] == 1 and len(calls) == 1

    for status in (2, 3):
        calls.clear()
        monkeypatch.setattr(subprocess, "run", fake_run([_Done(status, "")]))
        assert bench.supervise([]) == status and len(calls) == 1
    capsys.readouterr()

    calls.clear()
    monkeypatch.setattr(subprocess, "run", fake_run([_Done(-6, ""), _Done(1, "")]))
    assert bench.supervise([]) == 1 and len(calls) == 2
    assert capsys.readouterr().out == ""
