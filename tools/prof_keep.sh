#!/bin/bash
# kernel-trace stats of the standalone headline forward (tools/bench_kernels.py keep); repo root, GPU box
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_keep -o k -- python3 $R/tools/bench_kernels.py keep > $R/gpurun_out/prof_keep.log 2>&1
cd $R
f=$(find gpurun_out/prof_keep -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -4 "$f" | cut -c1-200
find gpurun_out/prof_keep -name "*kernel_trace.csv" -delete
