#!/bin/bash
# rocprofv3 evidence for BASELINE configs 4 and 5 at their stated size (tools/run_config.py 4 | 5):
#   stats: kernel trace + per-kernel stats;  busy: GRBM_GUI_ACTIVE + SQ_VALU_MFMA_BUSY_CYCLES + SQ_BUSY_CU_CYCLES
# usage (repo root, GPU box): bash tools/profile_configs.sh 4|5
set -u
W=${1:-4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_c$W
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/tools/run_config.py $W 2 1 > $O/stats.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $O/busy -o b -- python3 $R/tools/run_config.py $W 1 1 > $O/busy.log 2>&1
cd $R
tail -5 $O/stats.log
find $O -name "*kernel_stats.csv" | head -2
