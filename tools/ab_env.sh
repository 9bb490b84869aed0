#!/bin/bash
# In-box A/B of the headline step under an environment switch: VAR=a against VAR=b, interleaved, K rounds
# (ms_per_step of bench.py --headline-only), then one rocprofv3 kernel-stats pass each.
# usage (GPU box, repo root): bash tools/ab_env.sh VAR a b [rounds] [kernel-name filter for the stats]
R=${GRAFT_REPO_ROOT:-$(pwd)}
V=$1; A=$2; B=$3; K=${4:-3}; F=${5:-whiten|proj64}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', round(d['ms_per_step'],4))"; }
{
for i in $(seq $K); do
  env $V=$A python3 bench.py --headline-only --no-traffic 2>/dev/null | line "$V=$A"
  env $V=$B python3 bench.py --headline-only --no-traffic 2>/dev/null | line "$V=$B"
done
cd /tmp && export TMPDIR=/tmp
for x in $A $B; do
  export $V=$x
  rm -rf $O/abp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/abp -o p -- python3 $R/bench.py --headline-only --no-traffic --blocks 1 --steps 10 --warmup 3 > /dev/null 2>&1
  echo "# $V=$x"
  python3 - "$(find $O/abp -name '*kernel_stats.csv' | head -1)" "$F" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print("  %-72s calls %4s  avg %9.1f us" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
rm -rf $O/abp
} > $O/ab_env_$V.txt 2>&1
cat $O/ab_env_$V.txt
