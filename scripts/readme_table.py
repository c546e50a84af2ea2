"""Rewrite the number rows of README.md's round-6 tables from one closing session's files (the rows are found by their first cell).
usage: python scripts/readme_table.py r06_final5 [profiles|gpurun_out]"""
import csv
import json
import os
import re
import sys

tag = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else "profiles"
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def line(name):
    return json.loads(open(os.path.join(root, src, "%s_bench_%s.json" % (tag, name))).read().strip().splitlines()[-1])


def trace_us(name, kernel):
    for r in csv.DictReader(open(os.path.join(root, src, "%s_kernel_stats%s.csv" % (tag, name)))):
        if kernel in r["Name"]:
            return float(r["AverageNs"]) / 1e3
    return None


a, b = line("20_5"), line("200_10")
liv, mul, street = line("livox100"), line("mulran"), line("street")
s2, s8, r2 = line("2streams_1gpu"), line("8streams_1gpu"), line("2rank_gloo")


def cpu(d):
    nat = d["cpu_baseline"].get("march_native", {}).get("value")
    return d["cpu_baseline"]["value"], d["speedup_vs_cpu"], nat


def pose(d):
    pe = d["pose_error_vs_cpu"]
    return "%.0e m, %.0e rad after %d frames" % (pe["translation_m"], pe["rotation_rad"], pe["after_frames"])


def gap(d):
    hs = d["host_side"]
    return 1e3 * hs["device_gap_ms"] / max(1, hs["frames"] - 1)


rows = {
    "| RegisterFrame, host input, frames queued |": "| RegisterFrame, host input, frames queued | **%.0f scans/s** (%.3f ms / frame) | **%.0f scans/s** (%.3f ms) | 3009 / 2825 |"
    % (a["value"], a["ms_per_step"], b["value"], b["ms_per_step"]),
    "| CPU baseline (the oracle,": "| CPU baseline (the oracle, best of 1/8/16/32 threads; 16 CPUs) | %.1f scans/s -> **%.1fx** (`-march=native`: %.1f scans/s) | %.1f -> **%.1fx** (%.1f scans/s) | 21.0x / 21.9x |"
    % (cpu(a) + cpu(b)),
    "| pose difference GPU vs CPU trajectory |": "| pose difference GPU vs CPU trajectory | %s | %s | 7e-15 / 8e-15 m |" % (pose(a), pose(b)),
    "| `k_icp` per launch (hipEvents": "| `k_icp` per launch (hipEvents / `rocprofv3` trace) / later iteration | %.1f / %.1f us; %.1f us | %.1f / %.1f us; %.1f us | 275 / 311 us; 10.9 / 11.6 us |"
    % (a["roofline"]["ms_per_launch"] * 1e3, trace_us("_20_5", "k_icp<false, false>"), a["icp_last_launch"]["later_iterations_us"],
       b["roofline"]["ms_per_launch"] * 1e3, trace_us("", "k_icp<false, false>"), b["icp_last_launch"]["later_iterations_us"]),
    "| `k_icp` roofline (algorithmic bytes": "| `k_icp` roofline (algorithmic bytes / hipEvent time) | %.0f GB/s = **%.1f %%** of 8 TB/s | %.0f GB/s = **%.1f %%**; PMC HBM traffic %.1f MB per launch | 13.1 %% / 22.4 %% |"
    % (a["roofline"]["achieved"], 100 * a["roofline"]["frac"], b["roofline"]["achieved"], 100 * b["roofline"]["frac"], b["roofline"]["traffic"] / 1e6),
    "| device time between two registrations |": "| device time between two registrations | %.0f us | %.0f us | 39 / 40 |" % (gap(a), gap(b)),
    "| `RegisterFrame` blocking with both clouds returned": "| `RegisterFrame` blocking with both clouds returned: ctypes / C++ | %.0f / %.0f | %.0f / %.0f | 876 / -- |"
    % (a["sync_with_outputs"]["scans_per_s"], a["sync_with_outputs_cpp"]["scans_per_s"], b["sync_with_outputs"]["scans_per_s"], b["sync_with_outputs_cpp"]["scans_per_s"]),
    "| 1M-point 128 x 8192 rays": "| 1M-point 128 x 8192 rays, voxel 0.1 m, 100 frames | **%.0f scans/s**, %.1fx CPU; `k_icp<.., true>` %.3f ms per launch (trace %.3f), **%.1f %%**; PMC %.0f MB per launch | 568, 32.7 %%, 1088 MB |"
    % (liv["value"], liv["speedup_vs_cpu"], liv["roofline"]["ms_per_launch"], trace_us("_livox100", "k_icp<false, true>") / 1e3, 100 * liv["roofline"]["frac"],
       liv["roofline"]["traffic"] / 1e6),
    "| MulRan-like OS1-64, deskew on |": "| MulRan-like OS1-64, deskew on | %.0f scans/s, %.1fx CPU | 3115 |" % (mul["value"], mul["speedup_vs_cpu"]),
    "| bare street |": "| bare street | %.0f scans/s, %.1fx CPU | 4699 |" % (street["value"], street["speedup_vs_cpu"]),
    "| 2 / 8 streams in-process on ONE GPU": "| 2 / 8 streams in-process on ONE GPU through `kicp_batch_*`; 2 ranks (gloo) stacked on GPU 0 -- plumbing runs | %.0f / %.0f; %.0f scans/s | 4105 / 3708; 3138 |"
    % (s2["value"], s8["value"], r2["value"]),
}
p = os.path.join(root, "README.md")
out, hit = [], set()
inside = False
for ln in open(p).read().split("\n"):
    if ln.startswith("## Round 6 result"):
        inside = True
    elif ln.startswith("## Round 5 result"):
        inside = False
    if inside:
        for k, v in rows.items():
            if ln.startswith(k):
                ln = v
                hit.add(k)
    out.append(ln)
open(p, "w").write("\n".join(out))
print("rows rewritten: %d of %d" % (len(hit), len(rows)), [k for k in rows if k not in hit])
print("20/5: %.0f scans/s, first %.1f later %.2f us; 200/10: %.0f, first %.1f later %.2f" % (
    a["value"], a["icp_last_launch"]["first_iteration_us"], a["icp_last_launch"]["later_iterations_us"], b["value"],
    b["icp_last_launch"]["first_iteration_us"], b["icp_last_launch"]["later_iterations_us"]))
