#!/bin/bash
# A/B of one environment knob inside the headline step (rocprofv3 kernel stats, one box, alternating):
#   bash tools/ab_env_kernel.sh VAR "v1 v2 ..." kernel-name-substring [substring2 ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06; mkdir -p $O
VAR=$1; VALS=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for v in $VALS; do
  export $VAR=$v
  rm -rf $O/abk
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/abk -o p -- python3 $R/bench.py --headline-only --blocks 1 --steps 8 --warmup 2 > $O/abk.log 2>&1
  f=$(find $O/abk -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$VAR=$v" "$@" <<'PY'
import csv, sys
subs = sys.argv[3:]
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    tot += float(r["TotalDurationNs"])
    if any(s in r["Name"] for s in subs):
        print("%s  %-70s avg %8.1f us x %s" % (sys.argv[2], r["Name"][:70], float(r["AverageNs"]) / 1e3, r["Calls"]))
print("%s  all kernels: %.1f us per step" % (sys.argv[2], tot / 1e3 / 10))
PY
done
rm -rf $O/abk
