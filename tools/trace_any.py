"""Per-kernel totals of a rocprofv3 --kernel-trace sqlite database (whole run).  usage: trace_any.py <db> [top]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.cursor().execute("select name, end-start from kernels"))
agg = {}
for n, d in rows:
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    if "at::native" in n:
        n = "torch-native"
    a = agg.setdefault(n, [0, 0])
    a[0] += d
    a[1] += 1
tot = sum(a[0] for a in agg.values())
for n, (t, c) in sorted(agg.items(), key=lambda x: -x[1][0])[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f"{t / 1e6:10.2f} ms {100 * t / tot:5.1f}%  {c:6d}  {n[:90]}")
