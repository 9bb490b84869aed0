#!/bin/bash
# In-box A/B of the bench: ab_base/ (a built copy of an older commit: git archive <rev> | tar -x -C ab_base) against
# the working tree, interleaved, K rounds.  Box-to-box variance is +-2-10 %, so only numbers from ONE box compare.
# usage (GPU box, repo root): bash tools/ab_bench.sh [rounds] [extra bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
K=${1:-3}
shift
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', round(d['ms_per_step'],3))"; }
for i in $(seq $K); do
  (cd $R/ab_base && python3 bench.py --no-cpu-baseline --no-s1 --no-graph "$@" 2>/dev/null | line base)
  (cd $R && python3 bench.py --no-cpu-baseline --no-s1 --no-graph "$@" 2>/dev/null | line new)
done
