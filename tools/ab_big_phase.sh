#!/bin/bash
# Round-4 verdict item 7 (i): the in-phase experiment on the large-M kernels (two 256-thread workgroups per CU).
# One box, one process per variant, rocprofv3 --kernel-trace --stats on the config-4-cut step (M = 500):
#   base | second workgroup started 2 / 4 / 8 x 256 cycles late | second workgroup at s_setprio 1 | both
# usage (GPU box, repo root): bash tools/ab_big_phase.sh [config]  -> gpurun_out/r05/r05_big_phase_ab.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-4cut}
O=$R/gpurun_out/r05
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
OUT=$O/r05_big_phase_ab_$CFG.txt
: > $OUT
for v in "0 0" "2 0" "4 0" "8 0" "0 1" "4 1" "0 0"; do
  set -- $v
  export GPSA_BIG_PHASE=$1 GPSA_BIG_PRIO=$2
  rm -rf $O/ph
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ph -o p -- python3 $R/tools/run_config.py $CFG 3 1 > $O/ph.log 2>&1
  echo "# GPSA_BIG_PHASE=$1 GPSA_BIG_PRIO=$2   $(grep -E 'ms per step|ms/step|step' $O/ph.log | tail -1)" >> $OUT
  f=$(find $O/ph -name "*kernel_stats.csv" | head -1)
  python3 - "$f" >> $OUT <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("big_quad", "gram_big_kernel", "big_accum_kernel", "prod_big")):
        print("  %-60s calls %4s  avg %10.1f us  total %10.1f ms" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
rm -rf $O/ph
cat $OUT
