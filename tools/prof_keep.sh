#!/bin/bash
# kernel-trace stats of one tools/bench_kernels.py case (default: keep); repo root, GPU box:  bash tools/prof_keep.sh elbo
W=${1:-keep}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$W -o k -- python3 $R/tools/bench_kernels.py $W > $R/gpurun_out/prof_$W.log 2>&1
cd $R
f=$(find gpurun_out/prof_$W -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -6 "$f" | cut -c1-160
find gpurun_out/prof_$W -name "*kernel_trace.csv" -delete
