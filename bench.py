#!/usr/bin/env python
"""bench.py -- RegisterFrame scans/s + ms/ICP-iteration on synthetic 64-beam ~130k-point scans.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one RegisterFrame (cpp/kiss_icp/pipeline/KissICP.cpp:35-68) of one scan: deskew/crop,
two voxel downsamples, the ICP loop against the local map, map update -- all on the GPU.

`value` is measured on the DROP-IN path: every scan starts in pageable HOST memory as a float64 (N,3)
numpy array (what the reference's dataloaders hand to register_frame, python/kiss_icp/pipeline.py:100-103)
and goes through kicp_pipeline_register_frame_async (include/kicp.h): staged into pinned memory, uploaded
under the previous frame's registration, queued.  The device-resident rate (scans already in HBM, which
the task statement names as the whole-job figure) and the fully synchronous RegisterFrame that also returns
its two clouds are reported beside it under their own keys (`device_resident`, `sync_with_outputs`).

One process per GPU; rank r runs its own synthetic sequence (weak scaling: BASELINE config 4, one stream
per GPU) and the ranks all-gather their poses over RCCL once per batch.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1]): KITTI-like HDL-64, 64 x 2048 = 131 072 rays per scan, voxel_size
1.0 m, max_range 100 m, no timestamps (python/kiss_icp/datasets/kitti.py:57), on the vegetated street scene
SURVEY.md section 8(d) specifies (source cloud 4-4.5 k points); 200 timed frames after 10 warm-up frames.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "kiss-icp_amd", "python")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="kitti", choices=["kitti", "kitti-street", "mulran", "livox"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0,
                    help="timed frames of the CPU baseline (0 = by workload: all K up to 40 full-size-voxel scans, 6 scans of the 1M-point configuration)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (device-resident, f32, synchronous)")
    ap.add_argument("--opt", action="append", default=[], help="name=value tuning option (kicp_set_option)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for plumbing tests)")
    ap.add_argument("--device", type=int, default=-1, help="force this device for every rank (plumbing tests on a 1-GPU box)")
    ap.add_argument("--gen-procs", type=int, default=0, help="processes generating the synthetic scans (0 = by core count)")
    return ap.parse_args()


def workload(name):
    """(dataset factory, its keyword arguments, KISSConfig overrides, description)"""
    from kiss_icp_amd.datasets import kitti_like, kitti_like_vegetated, livox_like, mulran_like

    if name == "kitti":
        return kitti_like_vegetated, {}, dict(deskew=False), "kitti-like HDL-64 64x2048 rays, vegetated street (SURVEY 8d), voxel 1.0 m"
    if name == "kitti-street":  # round 1's light scene
        return kitti_like, {}, dict(deskew=False), "kitti-like HDL-64 64x2048 rays, bare street, voxel 1.0 m"
    if name == "mulran":
        return mulran_like, {}, dict(deskew=True), "mulran-like OS1-64 64x1024 rays, deskew, voxel 1.0 m"
    return livox_like, {}, dict(deskew=False, voxel_size=0.1), "1M-pt 128x8192 rays, voxel 0.1 m"


def launch_plan(gpus, world_env, device, n_devices):
    """what `--gpus N` means for this process.  Returns (mode, devices, exchange):
      ("rank", None, None)         started by torch.distributed.run (WORLD_SIZE in the environment, what the driver does
                                   for N > 1): this process is ONE of N ranks, one stream on its own GPU;
      ("single", [d], None)        N = 1;
      ("in-process", devices, x)   N > 1 without a launcher: N streams driven from this process through the C-ABI's batch
                                   entry (kicp_batch_*: one worker thread per stream inside libkicp.so), poses all-gathered by
                                   RCCL called directly (x = "rccl") -- or, with --device d (every stream on the one GPU of a
                                   1-GPU box: plumbing), through a host communicator (x = "host"), because RCCL refuses two
                                   ranks on one device.
    Fewer visible devices than streams without --device is an error, never a silent single-stream run."""
    if gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if world_env is not None:
        if int(world_env) != gpus:
            raise SystemExit(f"--gpus {gpus} but WORLD_SIZE={world_env}: the launcher and the flag disagree")
        return "rank", None, None
    if gpus == 1:
        return "single", [device if device >= 0 else 0], None
    if device >= 0:
        return "in-process", [device] * gpus, "host"
    if n_devices < gpus:
        raise SystemExit(f"--gpus {gpus} needs {gpus} GPUs, {n_devices} visible (use --device D to stack the streams on one GPU "
                         "for a plumbing run)")
    return "in-process", list(range(gpus)), "rccl"


def visible_devices():
    """GPUs this box offers, asked in a child process: this one must not touch a HIP runtime before the scans exist
    (the generator pool forks) nor before torch has loaded its own (libkicp binds to the one already in the process)"""
    import subprocess

    code = "import sys; sys.path.insert(0, %r); from kiss_icp_amd import _cabi; print(_cabi.device_count())" % os.path.join(ROOT, "kiss-icp_amd", "python")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    try:
        return int(r.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        return 0


def usable_cpus():
    """CPUs this process may really use: the affinity mask capped by the cgroup's quota (a GPU box's container: 16 of 256)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, -(-int(quota) // int(period)))
    except Exception:  # noqa: BLE001 -- no cgroup v2 file: the affinity count stands
        pass
    return max(1, n)


def pmc_traffic(name):
    """HBM bytes per k_icp launch from the PMC counters (FETCH_SIZE + WRITE_SIZE, separate rocprofv3
    passes over this same command, corrected as calibrated on a known-size copy; scripts/pmc_to_json.py
    writes the file, scripts/gpu_call.sh (WHAT=pmc) collects the counters).  Counters cannot be collected inside the
    timed run, so this is the committed measurement of a SEPARATE profiled run -- or None."""
    try:
        doc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        return float(doc[name]["kernels"]["k_icp"]["hbm_bytes_per_launch"])
    except Exception:
        return None


def cpu_baseline(scans, warmup, steps, cfg, thread_counts=None):
    """the oracle (CPU restatement of the reference path; the upstream binary cannot be built
    here) timed on this box's host cores on the first `steps` timed frames.  Reported, never the target."""
    from oracle import oracle as O

    cores = O.num_procs()
    best = None
    detail = {}
    for threads in thread_counts or sorted({1, min(8, cores), min(16, cores), min(32, cores)}):
        kw = dict(cfg)
        kw["deskew"] = int(kw.get("deskew", False))
        k = O.KissICP(max_num_threads=threads, **kw)
        for i in range(warmup):
            k.register_frame_noout(scans[i][0], scans[i][1])
        iters = 0
        t0 = time.perf_counter()
        for i in range(warmup, warmup + steps):
            k.register_frame_noout(scans[i][0], scans[i][1])
            iters += k.last_stats()["iterations"]
        dt = time.perf_counter() - t0
        detail[threads] = {"scans_per_s": steps / dt, "ms_per_icp_iter": 1e3 * dt / max(1, iters), "pose": k.last_pose}
        if best is None or steps / dt > detail[best]["scans_per_s"]:
            best = threads
    return best, detail


def cpu_baseline_native(scans, warmup, steps, cfg, threads):
    """the same sample on the oracle built with -march=native for THIS box's host (SURVEY.md section 8d names the flag; the
    reference's own Release build is generic).  Compiled here, at run time, into a temporary directory -- a library built for
    the build container's CPU may not run on this one -- and timed in a child process (the checker's library is already
    loaded in this one).  Returns {"value", "flags"} or {"error"}: a reported figure, never a reason for the bench to fail."""
    import pickle
    import shutil
    import subprocess
    import tempfile

    tmp = tempfile.mkdtemp(prefix="kicp_native_")
    try:
        lib = os.path.join(tmp, "libkiss_oracle_native.so")
        flags = ["-O3", "-march=native", "-std=c11", "-fopenmp", "-ffp-contract=off", "-fPIC"]
        subprocess.run(["gcc"] + flags + ["-shared", "-o", lib, os.path.join(ROOT, "oracle", "kiss_oracle.c"), "-lm"],
                       check=True, capture_output=True, timeout=120)
        kw = dict(cfg)
        kw["deskew"] = int(kw.get("deskew", False))
        with open(os.path.join(tmp, "job.pkl"), "wb") as f:
            pickle.dump({"scans": [(a, b) for a, b in scans[:warmup + steps]], "warmup": warmup, "steps": steps, "threads": threads, "cfg": kw}, f)
        child = (
            "import pickle, sys, time, json\n"
            "sys.path.insert(0, %r)\n"
            "from oracle import oracle as O\n"
            "j = pickle.load(open(%r, 'rb'))\n"
            "k = O.KissICP(max_num_threads=j['threads'], **j['cfg'])\n"
            "for i in range(j['warmup']): k.register_frame_noout(*j['scans'][i])\n"
            "t0 = time.perf_counter()\n"
            "for i in range(j['warmup'], j['warmup'] + j['steps']): k.register_frame_noout(*j['scans'][i])\n"
            "print(json.dumps({'scans_per_s': j['steps'] / (time.perf_counter() - t0)}))\n" % (ROOT, os.path.join(tmp, "job.pkl")))
        env = dict(os.environ, KISS_ORACLE_LIB=lib)
        r = subprocess.run([sys.executable, "-c", child], check=True, capture_output=True, text=True, timeout=600, env=env)
        return {"value": json.loads(r.stdout.strip().splitlines()[-1])["scans_per_s"], "cores": threads, "flags": " ".join(flags)}
    except Exception as e:  # noqa: BLE001
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def host_communicator(n_ranks, device):
    """a kicp_batch_comm that moves the blocks through host memory with the C-ABI's own device copies: for N streams
    stacked on ONE device (a plumbing run on a 1-GPU box), where RCCL cannot be used"""
    import ctypes as C
    import threading

    from kiss_icp_amd import _cabi

    L = _cabi.lib()
    barrier = threading.Barrier(n_ranks)
    blocks = [None] * n_ranks

    def all_gather(ctx, rank, d_send, d_recv, nbytes, stream):
        try:
            if L.kicp_device_synchronize(device):
                return 2
            mine = (C.c_ubyte * nbytes)()
            if L.kicp_device_download(device, mine, d_send, nbytes):
                return 2
            blocks[rank] = bytes(mine)
            barrier.wait(timeout=120)
            joined = b"".join(blocks)
            if L.kicp_device_upload(device, d_recv, joined, len(joined)):
                return 2
            barrier.wait(timeout=120)
            return 0
        except Exception:  # noqa: BLE001 -- nothing may propagate into the C caller
            return 2

    return _cabi.BatchComm(None, _cabi.BatchComm.INIT(0), _cabi.BatchComm.ALL_GATHER(all_gather), _cabi.BatchComm.FINALIZE(0))


def main_in_process(args, devices, exchange):
    """N > 1 streams from ONE process: the C-ABI's batch entry (worker thread per stream inside the library, each bound to
    its GPU), poses all-gathered once per batch by RCCL called directly.  No Python and no torch in the per-frame path
    beyond handing the N host arrays of a round to kicp_batch_register_frames."""
    import numpy as np

    from kiss_icp_amd import multistream
    from kiss_icp_amd.datasets import generate_scans

    W, K, S = args.warmup, args.steps, len(devices)
    factory, ds_kw, cfg_over, workload_name = workload(args.workload)
    t_gen = time.perf_counter()
    streams = []
    for r in range(S):
        sc = generate_scans(factory, dict(ds_kw, seed=multistream.stream_seed(args.seed, r), n_frames=W + K), range(W + K),
                            processes=args.gen_procs or None)
        streams.append([(np.ascontiguousarray(p, dtype=np.float64), np.ascontiguousarray(t, dtype=np.float64)) for p, t in sc])
    t_gen = time.perf_counter() - t_gen

    import torch  # first: libkicp must bind to the HIP runtime torch has already loaded

    from kiss_icp_amd import _cabi
    from kiss_icp_amd.config import load_config

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    for kv in args.opt:
        name, value = kv.split("=")
        _cabi.set_option(name, int(value))
    comm = host_communicator(S, devices[0]) if exchange == "host" else None
    batch = multistream.StreamBatch(load_config(**cfg_over), devices, comm=comm)

    def drive(lo, hi):
        for f in range(lo, hi):
            batch.register_frames([streams[r][f][0] for r in range(S)], [streams[r][f][1] for r in range(S)])
        batch.sync()

    def sync_devices():
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)

    drive(0, W)  # untimed warm-up
    sync_devices()
    cpu0 = time.process_time()
    t0 = time.perf_counter()
    drive(W, W + K)
    sync_devices()
    elapsed = time.perf_counter() - t0
    cpu_s = time.process_time() - cpu0
    poses = [batch.poses(r) for r in range(S)]
    assert all(len(p) == K for p in poses), [len(p) for p in poses]
    out = {
        "metric": "RegisterFrame scans/s", "value": S * K / elapsed, "unit": "scans/s", "n_gpus": S, "steps": K, "warmup": W,
        "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": workload_name, "input": "host float64 arrays (pageable) -> kicp_batch_register_frames",
            "streams": S, "parallelism": f"streams{S}", "devices": devices,
            "launcher": "in-process: kicp_batch_* (one worker thread per stream inside libkicp.so)",
            "exchange": "ncclAllGather called directly, once per batch" if exchange == "rccl"
                        else "host communicator (streams stacked on one device: plumbing, not a scaling measurement)",
            "frames_driven": W + K,
        },
        "rccl_ranks": S if exchange == "rccl" else 0,
        # what the HOST spent on the timed frames (all threads of this process: callers, staging helpers, batch workers, the
        # runtime's own) against the CPUs it may use -- whether N streams fit a box's cores is decided here, not on the GPUs
        "host_cpu": {"process_cpu_s": cpu_s, "cpu_s_per_frame": cpu_s / (S * K), "cpu_busy": cpu_s / elapsed, "usable_cpus": usable_cpus()},
        "pose_gather_s": batch.gather_seconds(),
        "scan_generation_s": t_gen,
    }
    batch.close()
    print(json.dumps(out))


def main():
    args = parse_args()
    W, K = args.warmup, args.steps
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        mode, devices, exchange = launch_plan(args.gpus, None, args.device, visible_devices())
        return main_in_process(args, devices, exchange)
    launch_plan(args.gpus, os.environ.get("WORLD_SIZE"), args.device, 1 << 30)  # the launcher and the flag must agree
    rank = int(os.environ.get("RANK", "0"))
    factory, ds_kw, cfg_over, workload_name = workload(args.workload)
    # the synthetic scans first, by a pool of processes, before this process touches the GPU runtime
    from kiss_icp_amd import multistream
    from kiss_icp_amd.datasets import generate_scans

    t_gen = time.perf_counter()
    scans = generate_scans(factory, dict(ds_kw, seed=multistream.stream_seed(args.seed, rank), n_frames=W + K), range(W + K),
                           processes=args.gen_procs or None)
    t_gen = time.perf_counter() - t_gen

    import numpy as np
    import torch  # first: libkicp must bind to the HIP runtime torch has already loaded

    from kiss_icp_amd import _cabi
    from kiss_icp_amd.config import load_config
    from kiss_icp_amd.kiss_icp import KissICP

    rank, local_rank, world = multistream.dist_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    if args.device >= 0:
        local_rank = args.device
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = multistream.init_process_group(args.backend) if world > 1 else None
    comm_device = device if args.backend == "nccl" else None  # gloo exchanges host tensors
    if args.device >= 0 and world > 1:
        # ranks stacked on ONE GPU (a plumbing run): the persistent registrations of different PROCESSES are ordered by nobody
        # (the launch gate is per process), so each rank takes 1 / world of the co-resident workgroups and all of them fit
        # side by side -- with full grids two ranks wait for each other's workgroups until one gives up (KICP_ERR_TIMEOUT)
        _cabi.set_option("icp_device_streams", min(world, 8))
    for kv in args.opt:
        name, value = kv.split("=")
        _cabi.set_option(name, int(value))

    # what the reference's dataloaders yield: float64 (N,3) arrays in pageable host memory
    host = [(np.ascontiguousarray(p, dtype=np.float64), np.ascontiguousarray(t, dtype=np.float64)) for p, t in scans]

    pipe = KissICP(load_config(**cfg_over), device_id=local_rank)
    multistream.run_batch_host(pipe, host[:W], dist, comm_device)  # W untimed warm-up frames
    pipe.icp_timing(reset=True)
    pipe.host_stats(reset=True)

    # ---- timed region: exactly K frames, barrier + synchronize on both sides -------------------
    multistream.barrier(dist)
    torch.cuda.synchronize()
    cpu0 = time.process_time()
    t0 = time.perf_counter()
    local_poses, all_poses = multistream.run_batch_host(pipe, host[W:W + K], dist, comm_device)
    torch.cuda.synchronize()
    multistream.barrier(dist)
    elapsed = time.perf_counter() - t0
    cpu_s = time.process_time() - cpu0
    elapsed = multistream.max_over_ranks(elapsed, dist, comm_device)
    icp = pipe.icp_timing()
    stats = pipe.last_stats()
    host_side = pipe.host_stats()

    if dist is not None:  # every rank leaves the process group together, before rank 0 goes on to report
        try:
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001 -- a teardown problem must not cost the measurement
            pass
    if rank != 0:
        return

    out = {
        "metric": "RegisterFrame scans/s",
        "value": world * K / elapsed,
        "unit": "scans/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": 1e3 * elapsed / K,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": workload_name,
            "input": "host float64 arrays (pageable) -> kicp_pipeline_register_frame_async",
            "streams": world,
            "parallelism": f"streams{world}" if world > 1 else "single-stream",
            "frames_driven": W + K,
            "n_raw": int(stats["n_raw"]),
            "n_frame_downsample": int(stats["n_frame_downsample"]),
            "n_source": int(stats["n_source"]),
            "map_voxels": int(stats["map_voxels"]),
            "icp_iters_per_frame": icp["iterations"] / max(1, icp["launches"]),
            "icp_workgroups": pipe.icp_profile()["workgroups"],
        },
        "ms_per_icp_iter": icp["total_ms"] / max(1, icp["iterations"]),
        # how many ranks the pose exchange of this run really spanned, and through what: a driver can tell from the line alone
        # whether RCCL saw N ranks (backend "nccl" IS RCCL on ROCm), a plumbing backend did, or there was nothing to exchange
        "rccl_ranks": world if (world > 1 and args.backend == "nccl") else 0,
        "exchange": {"ranks": world, "backend": args.backend if world > 1 else None,
                     "collective": "all_gather of the new poses, once per batch of frames" if world > 1 else None,
                     "launcher": "torch.distributed.run, one rank per GPU" if "WORLD_SIZE" in os.environ else "single process"},
        # what the host side of the K timed calls did (kicp_pipeline_host_stats): waits must be zero in steady state
        "host_side": host_side,
        # this rank's host cost of the timed frames (all threads of the process) against the CPUs it may use: N ranks on a box
        # share them, so cpu_busy x N against usable_cpus says whether the hosts or the GPUs bound an N-GPU run
        "host_cpu": {"process_cpu_s": cpu_s, "cpu_s_per_frame": cpu_s / K, "cpu_busy": cpu_s / elapsed, "usable_cpus": usable_cpus()},
        "scan_generation_s": t_gen,
    }
    cyc, tk = pipe.icp_clock()
    out["icp_shader_mhz"] = 100.0 * cyc / max(1, tk)
    first_us, total_us, n_it = pipe.icp_first_iteration()  # last frame of the timed region
    out["icp_last_launch"] = {"first_iteration_us": first_us, "total_us": total_us, "iterations": n_it,
                              "later_iterations_us": (total_us - first_us) / max(1, n_it - 1)}
    # roofline of the dominant kernel (k_icp): algorithmic bytes of AlignPointsToMap
    # (per iteration N_src*(24+24) + N_src*27*16 + E*24 + 336) / device time
    # measured with hipEvents on the pipeline's own stream around every k_icp launch of the timed region.
    if icp["total_ms"] > 0:
        achieved = icp["algorithmic_bytes"] / (icp["total_ms"] * 1e-3) / 1e9
        out["roofline"] = {
            "bound": "hbm", "kernel": "k_icp", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(args.workload),
            "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this workload -- kitti: 60 + 10 frames, livox: 30 + 4 --, corrected by a 1 GiB calibration copy on the same box; null = not collected for this workload)",
            "bytes_per_launch": icp["algorithmic_bytes"] / max(1, icp["launches"]),
            "ms_per_launch": icp["total_ms"] / max(1, icp["launches"]),
        }

    # ---- secondary measurements, outside the timed region: the same K frames on fresh pipelines ------------
    if not args.no_extras and world == 1:
        def drive(fn_enqueue, prepare=None):
            k = KissICP(load_config(**cfg_over), device_id=local_rank)
            for f in host[:W]:
                k.register_frame_async(*f)
            k.sync()
            items = prepare() if prepare else host[W:W + K]
            k.host_stats(reset=True)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for it in items:
                fn_enqueue(k, it)
            k.sync()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            return k, K / dt

        # (a) scans already resident in HBM (float64), queued back-to-back
        dev_pts = [torch.from_numpy(p).to(device) for p, _ in host[W:W + K]]
        dev_ts = [torch.from_numpy(t).to(device) if len(t) else None for _, t in host[W:W + K]]
        frames = [(d.data_ptr(), d.shape[0], t.data_ptr() if t is not None else None, t.shape[0] if t is not None else 0)
                  for d, t in zip(dev_pts, dev_ts)]
        kd, rate_dev = drive(lambda k, f: k.register_frame_device(*f), lambda: frames)
        hs_dev = kd.host_stats()
        out["device_resident"] = {"scans_per_s": rate_dev, "ms_per_frame": 1e3 / rate_dev, "icp_workgroups": kd.icp_profile()["workgroups"],
                                  "same_trajectory_as_host_input": bool((kd.last_pose == local_poses[-1]).all()),
                                  # where its time went: device time between registrations (mean / worst single gap), host-side waits
                                  "device_gap_ms_per_frame": hs_dev["device_gap_ms"] / K, "max_device_gap_ms": hs_dev["max_device_gap_ms"],
                                  "map_grows": hs_dev["map_grows"], "map_rehashes": hs_dev["map_rehashes"], "buffer_grows": hs_dev["buffer_grows"],
                                  "counter_refreshes": hs_dev["counter_refreshes"], "wait_ms": hs_dev["wait_ms"]}
        del dev_pts, dev_ts
        # (b) the sensor's native float32 points in host memory (no widening, no narrowing)
        host32 = [(p.astype(np.float32), t) for p, t in host[W:W + K]]
        k32, rate32 = drive(lambda k, f: k.register_frame_async(*f), lambda: host32)
        out["host_float32_input"] = {"scans_per_s": rate32, "ms_per_frame": 1e3 / rate32,
                                     "same_trajectory_as_host_input": bool((k32.last_pose == local_poses[-1]).all())}
        # (c) fully synchronous RegisterFrame: wait for the pose after every frame, without and with the two
        #     clouds the reference's RegisterFrame returns (KissICP.cpp:67) copied back to host arrays
        ks = KissICP(load_config(**cfg_over), device_id=local_rank)
        for f in host[:W]:
            ks.register_frame_async(*f)
        ks.sync()
        t = time.perf_counter()
        for f in host[W:W + K]:
            ks.register_frame_async(*f)
            ks.sync()
        rate_sync = K / (time.perf_counter() - t)
        ko = KissICP(load_config(**cfg_over), device_id=local_rank)
        for f in host[:W]:
            ko.register_frame(*f)  # (warmed up through the entry that is timed: until round 6 the asynchronous entry warmed it up, and
            # the first timed call paid for the third stream, the pinned output buffer and the first touch of the result arrays --
            # half of the 20-frame figure)
        t = time.perf_counter()
        for f in host[W:W + K]:
            ko.register_frame(*f)  # returns (preprocessed frame, source) as numpy arrays
        rate_out = K / (time.perf_counter() - t)
        # (d) the same signature in C++: kiss_icp::pipeline::KissICP::RegisterFrame (KissICP.cpp:35-68; what
        #     ros/src/OdometryServer.cpp:162 calls) returning its two std::vector<Eigen::Vector3d>, driven through the pybind
        #     module's _KissICP (a thin call: the scan is passed as a view, the result tuple is not converted)
        try:
            from kiss_icp_amd.metrics import _pybind

            pm = _pybind()
            pc, kc = load_config(**cfg_over), pm._KISSConfig()
            for section in (pc.data, pc.mapping, pc.registration, pc.adaptive_threshold):
                for name, v in vars(section).items():
                    if v is not None and hasattr(kc, name):
                        setattr(kc, name, v)
            kcpp = pm._KissICP(kc)
            for f in host[:W]:
                kcpp._register_frame(f[0], f[1])
            t = time.perf_counter()
            for f in host[W:W + K]:
                kcpp._register_frame(f[0], f[1])
            rate_cpp = K / (time.perf_counter() - t)
            out["sync_with_outputs_cpp"] = {"scans_per_s": rate_cpp, "ms_per_frame": 1e3 / rate_cpp,
                                            "same_trajectory_as_host_input": bool((np.asarray(kcpp._pose()) == local_poses[-1]).all())}
        except Exception as e:  # (the C++ layer is optional: make -C kiss-icp_amd/cpp)
            out["sync_with_outputs_cpp"] = {"error": repr(e)}
        out["sync_per_frame"] = {"scans_per_s": rate_sync, "ms_per_frame": 1e3 / rate_sync,
                                 "same_trajectory_as_host_input": bool((ks.last_pose == local_poses[-1]).all())}
        out["sync_with_outputs"] = {"scans_per_s": rate_out, "ms_per_frame": 1e3 / rate_out,
                                    "same_trajectory_as_host_input": bool((ko.last_pose == local_poses[-1]).all())}

    if world == 1 and not args.no_cpu_baseline:
        # a BOUNDED sample of the same workload: the first S timed frames (all of them for the driver's 20-frame run)
        big = args.workload == "livox"
        S = min(K, args.cpu_sample or (6 if big else 40))
        best, detail = cpu_baseline(scans, W, S, cfg_over, thread_counts=[16] if big else None)
        D = np.linalg.inv(detail[best]["pose"]) @ local_poses[S - 1]
        out["cpu_baseline"] = {
            "value": detail[best]["scans_per_s"], "unit": "scans/s", "cores": best, "kind": "port",
            "sample": f"the first {S} of the {K} timed frames (after the same {W} warm-up frames), host arrays",
            "what": "oracle/kiss_oracle.c -- a C restatement of the reference path, NOT the upstream binary (Eigen/Sophus/tsl/TBB "
                    "are not installed); OpenMP in the reference's three TBB sites; gcc -O3 -ffp-contract=off, no -march=native "
                    "(the reference's Release build is generic x86-64 too); march_native: the same sample and threads on a "
                    "-march=native build made on this box",
            "ms_per_icp_iter": detail[best]["ms_per_icp_iter"],
            "by_threads": {str(t): d["scans_per_s"] for t, d in detail.items()},
            "host_cores": os.cpu_count(),
        }
        if 1 in detail:
            out["cpu_baseline"]["single_thread_scans_per_s"] = detail[1]["scans_per_s"]
        # the same sample, same threads, on a -march=native build of the port made on this box (both factors in the line)
        out["cpu_baseline"]["march_native"] = cpu_baseline_native(scans, W, S, cfg_over, best)
        out["speedup_vs_cpu"] = out["value"] / detail[best]["scans_per_s"]
        out["pose_error_vs_cpu"] = {
            "translation_m": float(np.linalg.norm(D[:3, 3])),
            "rotation_rad": float(np.arccos(min(1.0, max(-1.0, (np.trace(D[:3, :3]) - 1.0) / 2.0)))),
            "after_frames": W + S,
        }
    print(json.dumps(out))


if __name__ == "__main__":
    # (measured in THIS process, once: a bench whose process dies has failed -- there is no second attempt)
    main()
