#!/bin/bash
# Ordered launch list of one S = 1 step (tools/step_timeline.py)  -> gpurun_out/tl/s1.timeline.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/tl
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/s1
rocprofv3 --kernel-trace --output-format csv -d $O/s1 -o t -- python3 $R/bench.py --headline-only --blocks 1 --steps 6 --warmup 3 --S 1 > $O/s1.log 2>&1
cd $R
f=$(find $O/s1 -name "*kernel_trace.csv" | head -1)
python3 tools/step_timeline.py $f > $O/s1.timeline.txt
cat $O/s1.timeline.txt
rm -rf $O/s1
