"""top kernels of a rocprofv3 --kernel-trace --stats run, per step:  python tools/stats_top.py <dir> <steps> [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
steps = int(sys.argv[2]); n = int(sys.argv[3]) if len(sys.argv) > 3 else 16
rows = list(csv.DictReader(open(f)))
print("sum per step (us):", round(sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e3, 1))
for r in rows[:n]:
    print("%8.1f us/step  calls/step %5.1f  %s" % (float(r["TotalDurationNs"]) / steps / 1e3, int(r["Calls"]) / steps, r["Name"][:100]))
