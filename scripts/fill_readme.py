"""Fill README.md's @PLACEHOLDERS@ of the current round's result section from one closing session's files.
usage: python scripts/fill_readme.py r06_final2 [gpurun_out|profiles]   (prints what it could not fill)"""
import csv
import json
import os
import re
import sys

tag = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else "profiles"
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def line(name):
    try:
        return json.loads(open(os.path.join(root, src, "%s_bench_%s.json" % (tag, name))).read().strip().splitlines()[-1])
    except Exception:
        return None


def trace_us(name, kernel):
    try:
        for r in csv.DictReader(open(os.path.join(root, src, "%s_kernel_stats%s.csv" % (tag, name)))):
            if kernel in r["Name"]:
                return float(r["AverageNs"]) / 1e3
    except Exception:
        pass
    return None


v = {}
for key, name in (("20", "20_5"), ("200", "200_10")):
    d = line(name)
    if not d:
        continue
    cb, rf, hs = d["cpu_baseline"], d["roofline"], d["host_side"]
    v["V" + key] = "%.0f" % d["value"]
    v["MS" + key] = "%.3f" % d["ms_per_step"]
    v["CPU" + key] = "%.1f" % cb["value"]
    v["X" + key] = "%.1f" % d["speedup_vs_cpu"]
    nat = cb.get("march_native", {})
    v["NAT" + key] = ("%.1f scans/s" % nat["value"]) if "value" in nat else "not measured"
    pe = d["pose_error_vs_cpu"]
    v["POSE" + key] = "%.0e m, %.0e rad after %d frames" % (pe["translation_m"], pe["rotation_rad"], pe["after_frames"])
    tr = trace_us("_20_5" if key == "20" else "", "k_icp<false, false>")
    v["K" + key] = "%.1f / %s" % (rf["ms_per_launch"] * 1e3, ("%.1f" % tr) if tr else "--")
    v["BW" + key] = "%.0f" % rf["achieved"]
    v["F" + key] = "%.1f" % (100 * rf["frac"])
    if rf.get("traffic"):
        v["PMC"] = "%.1f" % (rf["traffic"] / 1e6)
    v["GAP" + key] = "%.0f" % (1e3 * hs["device_gap_ms"] / max(1, hs["frames"] - 1))
    v["SYNC_OUT" + key] = "%.0f" % d["sync_with_outputs"]["scans_per_s"]
    v["SYNC_CPP" + key] = "%.0f" % d["sync_with_outputs_cpp"]["scans_per_s"]
    il = d["icp_last_launch"]
    v["LATER_" + ("YOUNG" if key == "20" else "STEADY")] = "%.1f" % il["later_iterations_us"]
    v["FIRST_" + ("YOUNG" if key == "20" else "STEADY")] = "%.0f" % il["first_iteration_us"]
if "SYNC_OUT20" in v and "SYNC_OUT200" in v:
    v["SYNC_OUT"] = "%s - %s" % tuple(sorted((v["SYNC_OUT20"], v["SYNC_OUT200"]), key=float))
    v["SYNC_CPP"] = "%s - %s" % tuple(sorted((v["SYNC_CPP20"], v["SYNC_CPP200"]), key=float))
    v["CPU"] = "%s - %s" % tuple(sorted((v["CPU20"], v["CPU200"]), key=float))
d = line("livox100")
if d:
    v["LIVOX_VALUE"] = "%.0f" % d["value"]
    v["LIVOX_X"] = "%.1f" % d["speedup_vs_cpu"]
    v["LIVOX_FRAC"] = "%.1f" % (100 * d["roofline"]["frac"])
    v["LIVOX_K"] = "%.3f" % d["roofline"]["ms_per_launch"]
    tr = trace_us("_livox100", "k_icp<false, true>")
    v["LIVOX_KT"] = ("%.3f" % (tr / 1e3)) if tr else "--"
    v["LIVOX_PMC"] = ("%.0f" % (d["roofline"]["traffic"] / 1e6)) if d["roofline"].get("traffic") else "--"
for key, name in (("MULRAN", "mulran"), ("STREET", "street")):
    d = line(name)
    if d:
        v[key] = "%.0f" % d["value"]
        v[key + "_X"] = "%.1f" % d["speedup_vs_cpu"]
for key, name in (("S2", "2streams_1gpu"), ("S8", "8streams_1gpu"), ("R2", "2rank_gloo")):
    d = line(name)
    if d:
        v[key] = "%.0f" % d["value"]
for a in sys.argv[3:]:  # free-text placeholders: NAME=text
    k, _, t = a.partition("=")
    v[k] = t
p = os.path.join(root, "README.md")
s = open(p).read()
s = re.sub(r"@([A-Z0-9_]+)@", lambda m: v.get(m.group(1), m.group(0)), s)
open(p, "w").write(s)
print("left:", sorted(set(re.findall(r"@([A-Z0-9_]+)@", s))))
