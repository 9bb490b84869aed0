"""Ordered launch list of ONE training step from a `rocprofv3 --kernel-trace --output-format csv` run of bench.py.

usage: step_timeline.py <kernel_trace.csv> [which step, default the middle one]
Steps are delimited by gpsa::adam_kernel.  Per launch: start offset, duration, idle gap before it, grid, name; then
the sums (busy, idle, wall) - the idle total is what the host / dependency stalls cost on top of the kernels.
"""
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                     int(r.get("Grid_Size_X", 0) or 0) // max(1, int(r.get("Workgroup_Size_X", 1) or 1)),
                     int(r.get("Grid_Size_Y", 1) or 1), int(r.get("Grid_Size_Z", 1) or 1),
                     int(r.get("Workgroup_Size_X", 0) or 0)))
rows.sort()
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
# (bench.py runs its blocks back to back: the 5th step sits inside the first timed block, whatever follows)
k = int(sys.argv[2]) if len(sys.argv) > 2 else min(5, len(ends) - 1)
a, b = ends[k - 1] + 1, ends[k] + 1
sel = rows[a:b]
t0 = sel[0][0]
busy = idle = 0
prev = rows[a - 1][1]
for s, e, n, gx, gy, gz, wg in sel:
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("gpsa::", "")
    gap = s - prev
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  gap {gap / 1e3:7.1f}  grid {gx:5d}x{gy}x{gz} wg {wg:4d}  {n[:100]}")
    busy += e - s
    idle += max(0, gap)
    prev = max(prev, e)
print(f"launches {len(sel)}  busy {busy / 1e3:.1f} us  idle {idle / 1e3:.1f} us  wall {(sel[-1][1] - rows[a - 1][1]) / 1e3:.1f} us")
