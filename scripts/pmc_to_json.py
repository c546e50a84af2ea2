"""Fold the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE summaries (scripts/pmc_summary.py output) of a
bench.py run into profiles/pmc_traffic.json, the file bench.py reads for roofline.traffic.

FETCH_SIZE / WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE counts 64 B per 128-B request of
a wide streaming read, i.e. half the bytes (MI355X_MICROARCH.md, HBM section); the calibration copy
of scripts/pmc_calibrate.py measured on the same box confirms both the unit and the factor and is
recorded alongside.  WRITE_SIZE needed no correction there."""
import json
import re
import sys

fetch_txt, write_txt, cal_fetch_txt, cal_write_txt, workload, out = sys.argv[1:7]


def parse(path):
    rows = {}
    for line in open(path):
        m = re.match(r"(\S+)\s+(.*?)\s+dispatches\s+(\d+)\s+mean\s+([\d.]+)\s+total\s+([\d.]+)", line)
        if m:
            rows[m.group(2).strip()] = {"dispatches": int(m.group(3)), "mean_kib": float(m.group(4))}
    return rows


fetch, write = parse(fetch_txt), parse(write_txt)
cal_f, cal_w = parse(cal_fetch_txt), parse(cal_write_txt)
copy_bytes = 1 << 30
cf = next(v for k, v in cal_f.items() if "copyBuffer" in k)["mean_kib"] * 1024
cw = next(v for k, v in cal_w.items() if "copyBuffer" in k)["mean_kib"] * 1024
fetch_factor, write_factor = copy_bytes / cf, copy_bytes / cw
try:
    doc = json.load(open(out))
except Exception:
    doc = {}
kern = {}
for name in sorted(set(fetch) | set(write)):
    if "kicp::" not in name:
        continue
    f = fetch.get(name, {}).get("mean_kib", 0.0) * 1024 * fetch_factor
    w = write.get(name, {}).get("mean_kib", 0.0) * 1024 * write_factor
    short = re.sub(r"^void\s+", "", name).replace("kicp::", "")
    short = re.sub(r"<.*", "", short)
    kern[short] = {"fetch_bytes_per_launch": f, "write_bytes_per_launch": w, "hbm_bytes_per_launch": f + w,
                   "dispatches": fetch.get(name, {}).get("dispatches", 0)}
doc[workload] = {
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --workload <w> --no-cpu-baseline --no-extras --steps 60 --warmup 10 (livox: 30 / 4)",
    "unit_note": "counters in KiB; factors from a 1 GiB device copy on the same box",
    "calibration": {"copy_bytes": copy_bytes, "fetch_reported_bytes": cf, "write_reported_bytes": cw,
                    "fetch_factor": fetch_factor, "write_factor": write_factor},
    "kernels": kern,
}
json.dump(doc, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(doc[workload]["kernels"].get("k_icp"), indent=1))
