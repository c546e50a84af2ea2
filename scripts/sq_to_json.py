"""Fold the SQ_* / GRBM counter passes of one kernel (scripts/pmc_summary.py outputs) into one JSON with the derived shares.
usage: sq_to_json.py KERNEL_SUBSTRING out.json pass1.txt pass2.txt ...
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave; SQ_INSTS_* count
wave-instructions; GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import json
import re
import sys

kernel, out = sys.argv[1], sys.argv[2]
c, disp = {}, 0
for path in sys.argv[3:]:
    for line in open(path):
        m = re.match(r"(\S+)\s+(.*?)\s+dispatches\s+(\d+)\s+mean\s+([\d.]+)", line)
        if m and kernel in m.group(2):
            c[m.group(1)] = float(m.group(4))
            disp = int(m.group(3))
d = {"kernel": kernel, "dispatches": disp, "per_launch_mean": c, "derived": {}}
g = d["derived"]
wc = c.get("SQ_WAVE_CYCLES")
if wc:
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS",
              "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS"):
        if k in c:
            g[k.lower() + "_share_of_wave_cycles"] = round(c[k] / wc, 4)
if "SQ_WAVES" in c:
    w = c["SQ_WAVES"]
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"):
        if k in c:
            g[k.lower() + "_per_wave"] = round(c[k] / w, 1)
    if wc:
        g["cycles_per_wave"] = round(4.0 * wc / w, 0)
if c.get("SQ_LDS_IDX_ACTIVE"):
    g["lds_bank_conflict_share_of_lds_active"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
if "GRBM_GUI_ACTIVE" in c:
    g["gpu_cycles_per_launch"] = round(c["GRBM_GUI_ACTIVE"] / 8.0, 0)
json.dump(d, open(out, "w"), indent=1)
print(json.dumps(g, indent=1))
