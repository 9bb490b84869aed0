#!/bin/bash
# Round-5 verdict item 6: one wait + barrier per TWO chunks in the large-M Gram kernel (gram_big_kernel_t<true>: four
# one-chunk slots, GPSA_BIG_PAIR=1) against one per chunk (three slots).  One box, one process per variant, alternating,
# rocprofv3 --kernel-trace --stats on the config-4-cut (M = 500) and config-5-cut (M = 1000) steps.
# usage (GPU box, repo root): bash tools/ab_big_pair.sh  -> gpurun_out/r06/r06_big_pair_ab.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
OUT=$O/r06_big_pair_ab.txt
: > $OUT
for CFG in 4cut 5cut; do
  for v in 0 1 0 1; do
    export GPSA_BIG_PAIR=$v
    rm -rf $O/pp
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/pp -o p -- python3 $R/tools/run_config.py $CFG 3 1 > $O/pp.log 2>&1
    echo "# config $CFG  GPSA_BIG_PAIR=$v   $(grep -E 'ms per step|ms/step|step' $O/pp.log | tail -1)" >> $OUT
    f=$(find $O/pp -name "*kernel_stats.csv" | head -1)
    python3 - "$f" >> $OUT <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("big_quad", "gram_big_kernel", "big_accum_kernel")):
        print("  %-60s calls %4s  avg %10.1f us  total %10.1f ms" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  done
done
rm -rf $O/pp
cat $OUT
