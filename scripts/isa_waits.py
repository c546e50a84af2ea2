"""Where does a wave of k_icp wait?  Compiles kicp_icp.hip for gfx950 to assembly with line tables (no GPU needed) and counts, per
source line of kicp_icp.hip that the instruction was inlined into, the FULL waits on the LDS / scalar-memory counter
(`s_waitcnt lgkmcnt(0)`: the wave stands until every outstanding LDS access has come back -- a dependent round trip) and on the
vector-memory counter (`vmcnt(0)`), the barriers and the instructions.  Round 6 used exactly this reading of the code object to
find the chains of dependent LDS round trips on one wave's critical path.

usage: python scripts/isa_waits.py [--rev GITREV] [--kernel 'k_icpILb0ELb0E'] [--lines A-B ...] [--top N]
  --rev    the sources of that revision (git archive into a scratch directory) instead of the working tree's
  --lines  sum over these line ranges of kicp_icp.hip as well (e.g. the phases of the iteration loop)"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
ap = argparse.ArgumentParser()
ap.add_argument("--rev")
ap.add_argument("--kernel", default="k_icpILb0ELb0E")
ap.add_argument("--lines", nargs="*", default=[])
ap.add_argument("--top", type=int, default=25)
args = ap.parse_args()

tmp = tempfile.mkdtemp(prefix="kicp_isa_")
src = os.path.join(ROOT, "kiss-icp_amd", "csrc")
if args.rev:
    subprocess.run("git -C %s archive %s kiss-icp_amd/csrc include | tar -x -C %s" % (ROOT, args.rev, tmp), shell=True, check=True)
    src = os.path.join(tmp, "kiss-icp_amd", "csrc")
asm = os.path.join(tmp, "kicp_icp.s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-gline-tables-only",
                "--cuda-device-only", "-S", "-o", asm, os.path.join(src, "kicp_icp.hip")], check=True, stderr=subprocess.DEVNULL)
per_line = collections.defaultdict(lambda: [0, 0, 0, 0])  # instructions, lgkm waits, vm waits, barriers
inside, where, total = False, None, [0, 0, 0, 0]
for line in open(asm):
    if not inside:
        inside = line.startswith("_ZN4kicp5" + args.kernel) and line.rstrip().endswith(":") or (line.startswith("_ZN4kicp5" + args.kernel) and ": " in line)
        continue
    if line.startswith(".Lfunc_end"):
        break
    m = re.match(r"\s+\.loc\s", line)
    if m:
        # the OUTERMOST frame of the inlined-at chain that lies in kicp_icp.hip: the line of the kernel the instruction belongs to
        hits = re.findall(r"kicp_icp\.hip:(\d+):", line)
        where = int(hits[-1]) if hits else None
        continue
    t = line.strip()
    if not t or t.startswith((";", ".")) or t.endswith(":"):
        continue
    rec = per_line[where]
    rec[0] += 1
    total[0] += 1
    if t.startswith("s_waitcnt"):
        if "lgkmcnt(0)" in t:
            rec[1] += 1
            total[1] += 1
        if "vmcnt(0)" in t:
            rec[2] += 1
            total[2] += 1
    if t.startswith("s_barrier"):
        rec[3] += 1
        total[3] += 1
if not args.lines and args.kernel.endswith("ELb0E"):
    # the phases of the group form's iteration, by the markers in the source
    text = open(os.path.join(src, "kicp_icp.hip")).read().split("\n")
    def last(marker):
        hits = [i + 1 for i, l in enumerate(text) if marker in l]
        return hits[-1] if hits else None
    marks = [("A", last("// ---- A ---")), ("B0 + B", last("// ---- B0")), ("C", last("// ---- C ---")), ("reduction + exchange", last("// ---- workgroup reduction")),
             ("solve", last("// ---- waves 0..3")), (None, last("iterations = it + 1;"))]
    if all(m[1] for m in marks):
        args.lines = ["%d-%d" % (marks[i][1], marks[i + 1][1] - 1) for i in range(len(marks) - 1)]
        print("phases of the group form's iteration: " + ", ".join("%s = lines %s" % (marks[i][0], args.lines[i]) for i in range(len(args.lines))))
print("%s (%s): %d instructions, %d full LDS waits, %d full memory waits, %d barriers" % (args.kernel, args.rev or "working tree", *total))
for r in args.lines:
    a, b = (int(x) for x in r.split("-"))
    s = [sum(v[i] for k, v in per_line.items() if k is not None and a <= k <= b) for i in range(4)]
    print("  lines %5d - %5d: %6d instructions, %4d full LDS waits, %4d full memory waits, %3d barriers" % (a, b, *s))
print("  source lines of kicp_icp.hip with the most full LDS waits:")
for k, v in sorted(per_line.items(), key=lambda kv: -kv[1][1])[:args.top]:
    if v[1]:
        print("    line %-6s %5d instructions %4d LDS waits %3d memory waits %2d barriers" % (k, *v))
