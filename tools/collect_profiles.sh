#!/bin/bash
# Collect the rocprofv3 evidence of the default bench on the GPU box into gpurun_out/prof/ :
#   stats/  kernel trace + per-kernel stats     fetch/ write/  HBM-side PMC passes (one counter each)
#   busy/   GRBM_GUI_ACTIVE + SQ_VALU_MFMA_BUSY_CYCLES      bench_line.json  the plain bench line
# usage (from the repo root, on the GPU box):  bash tools/collect_profiles.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-graph --no-s1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $B --steps 10 --warmup 2 > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $B --steps 3 --warmup 1 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $B --steps 3 --warmup 1 > $O/write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/busy -o b -- python3 $B --steps 3 --warmup 1 > $O/busy.log 2>&1
cd $R
python3 bench.py > $O/bench_line.json 2> $O/bench_line.log
for k in 1 2 4 8; do python3 bench.py --no-cpu-baseline --emulate-shard $k 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('shard 1/$k  ms_per_step', round(d['ms_per_step'],3), ' graph', round(d['graph_replay']['ms_per_step'],3))"; done > $O/shards.txt
ls -la $O $O/stats | head -30
