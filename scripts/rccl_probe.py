#!/usr/bin/env python
"""RCCL on the one GPU a gpurun box has: does it initialise, and what does one pose exchange cost?
(a) the C-ABI's batch entry (kicp_batch_*, RCCL called directly from libkicp.so), single rank;
(b) torch.distributed backend "nccl" (= RCCL), world size 1 -- the path bench.py --gpus N uses.
Prints one JSON object.  Not a scaling measurement: one rank exchanges with itself."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kiss-icp_amd", "python"))

import numpy as np  # noqa: E402


def main():
    out = {}
    from kiss_icp_amd.config import load_config
    from kiss_icp_amd.datasets import kitti_like
    from kiss_icp_amd.multistream import StreamBatch

    t0 = time.perf_counter()
    b = StreamBatch(load_config(deskew=False), [0])
    out["cabi_batch_create_s"] = time.perf_counter() - t0
    ds = kitti_like(seed=3, n_frames=24, beams=32, azimuth_steps=600)
    for i in range(4):
        b.register_frames([ds[i][0]])
    b.sync()
    lat, wall = [], []
    for k in range(200):
        t = time.perf_counter()
        b.sync()  # nothing queued: the cost of the exchange alone (upload 8 KiB block, ncclAllGather, download, stream sync)
        wall.append(time.perf_counter() - t)
        lat.append(b.gather_seconds())
    out["cabi_batch_empty_sync_us"] = {"median": 1e6 * float(np.median(wall)), "p90": 1e6 * float(np.percentile(wall, 90))}
    out["cabi_batch_gather_us"] = {"median": 1e6 * float(np.median(lat)), "p90": 1e6 * float(np.percentile(lat, 90)),
                                   "block_bytes": 8 * (2 + 16 * 64)}
    per = []
    for i in range(4, 24, 4):
        for j in range(i, i + 4):
            b.register_frames([ds[j][0]])
        b.sync()
        per.append(b.gather_seconds())
    out["cabi_batch_gather_after_4_frames_us"] = 1e6 * float(np.median(per))
    b.close()

    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    t0 = time.perf_counter()
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    x = torch.zeros(16, dtype=torch.float64, device="cuda:0")
    outs = [torch.empty_like(x)]
    dist.all_gather(outs, x)
    torch.cuda.synchronize()
    out["torch_nccl_init_plus_first_all_gather_s"] = time.perf_counter() - t0
    ts = []
    for _ in range(200):
        t = time.perf_counter()
        dist.all_gather(outs, x)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    out["torch_nccl_all_gather_128B_us"] = {"median": 1e6 * float(np.median(ts)), "p90": 1e6 * float(np.percentile(ts, 90))}
    out["torch_nccl_version"] = list(torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None
    dist.destroy_process_group()
    out["note"] = "world size 1: RCCL initialises and runs the collective on the one GPU present; no scaling is implied"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
