"""Summarise a rocprofv3 --pmc counter_collection.csv for kernels matching a substring.
usage: python tools/pmc_kernel.py <counter_collection.csv> <kernel substring>"""
import collections, csv, sys
rows = csv.DictReader(open(sys.argv[1]))
acc = collections.defaultdict(list)
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:32s} {sum(v) / len(v):.4e}  (n={len(v)})")
