#!/bin/bash
# Ordered launch list of one step (tools/step_timeline.py) at K=1 and for rank 0's share of an 8-way sharding.
# usage (GPU box, repo root): bash tools/timeline.sh [tag]  -> gpurun_out/tl/{k1,s8}.timeline.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/tl
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="--headline-only --blocks 1 --steps 6 --warmup 3"
rocprofv3 --kernel-trace --output-format csv -d $O/s8 -o t -- python3 $R/bench.py $A --emulate-shard 8 > $O/s8.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/k1 -o t -- python3 $R/bench.py $A > $O/k1.log 2>&1
cd $R
for d in s8 k1; do
  f=$(find $O/$d -name "*kernel_trace.csv" | head -1)
  python3 tools/step_timeline.py $f > $O/$d.timeline.txt
  tail -1 $O/$d.timeline.txt
  rm -rf $O/$d
done
