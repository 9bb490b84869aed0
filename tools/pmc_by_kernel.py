"""Per-kernel means of every counter in a rocprofv3 --pmc counter_collection.csv, one line per (kernel, grid).
usage: python tools/pmc_by_kernel.py <counter_collection.csv> [substring]"""
import collections, csv, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    if sub in r["Kernel_Name"]:
        n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("gpsa::", "")
        key = (n[:70], r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for v in acc.values() for c in v})
print("kernel | grid | " + " | ".join(names))
for key, v in sorted(acc.items()):
    print(f"{key[0]:70s} {key[1]:>9s} " + " ".join(f"{sum(v[c]) / max(1, len(v[c])):12.4e}" for c in names))
