#!/bin/bash
# Round-6 evidence for profiles/ (GPU box, repo root):  bash tools/collect_r06.sh [tag]  -> gpurun_out/r06/
#   <tag>_default_bench_line.json        the plain `python bench.py` line (the reference's two-call loop)
#   <tag>_default_bench_kernel_stats.csv rocprofv3 --kernel-trace --stats of the same command (shorter run)
#   <tag>_pmc_fetch_write_summary.txt    HBM-side FETCH_SIZE / WRITE_SIZE, separate --pmc passes (profiles/pmc_summarize.py)
#   <tag>_pmc_clock_mfma_busy.txt        GRBM_GUI_ACTIVE + SQ_VALU_MFMA_BUSY_CYCLES per kernel
#   <tag>_step_timeline_{k1,shard8}.txt  ordered launch list of one step
#   <tag>_strong_scaling_emulation.txt   rank 0's share of a K-way row sharding on one GPU
#   <tag>_config3_kernel_stats.csv       BASELINE config 3 (bench.py --workload 3)
set -u
T=${1:-r06_a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
python3 bench.py > $O/${T}_default_bench_line.json 2> $O/${T}_bench.log
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --headline-only --blocks 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $B --steps 10 --warmup 2 > $O/stats.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/${T}_default_bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $B --steps 3 --warmup 1 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $B --steps 3 --warmup 1 > $O/write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/busy -o b -- python3 $B --steps 3 --warmup 1 > $O/busy.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o s -- python3 $R/bench.py --workload 3 --steps 6 --warmup 2 --blocks 1 > $O/c3.log 2>&1
cp $(find $O/c3 -name "*kernel_stats.csv" | head -1) $O/${T}_config3_kernel_stats.csv
cd $R
python3 profiles/pmc_summarize.py $O/fetch $O/write > $O/${T}_pmc_fetch_write_summary.txt 2>&1
python3 tools/pmc_by_kernel.py $(find $O/busy -name "*counter_collection.csv" | head -1) > $O/${T}_pmc_clock_mfma_busy.txt 2>&1
bash tools/timeline.sh > /dev/null 2>&1
cp gpurun_out/tl/k1.timeline.txt $O/${T}_step_timeline_k1.txt
cp gpurun_out/tl/s8.timeline.txt $O/${T}_step_timeline_shard8.txt
for k in 1 2 4 8; do python3 bench.py --no-cpu-baseline --no-extras --no-graph --blocks 3 --emulate-shard $k 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('shard 1/$k  ms_per_step', round(d['ms_per_step'],3), ' S=1', round(d['secondary_S1']['ms_per_step'],3))"; done > $O/${T}_strong_scaling_emulation.txt
cp profiles/pmc_traffic.json $O/${T}_pmc_traffic.json
# rank 0's share of an 8-way sharding: KL terms at weight 1/8 on every rank (rounds 1-5) against owner computes (round 6)
bash tools/ab_shard8.sh 8 > $O/${T}_shard8_kl_share_ab.txt 2>&1
{
  echo "# full step with Adam, one MI355X (tools/run_config.py; config 3: bench.py --workload 3 above)"
  for w in 4cut 5cut; do python3 tools/run_config.py $w 6 3 2>/dev/null | tail -1; done
  python3 tools/run_config.py 4 3 2 2>/dev/null | tail -4
  python3 tools/run_config.py 5 2 1 2>/dev/null | tail -4
  GPSA_MICROBATCHES=8 python3 tools/run_config.py 5 3 2 2>/dev/null | tail -1
} > $O/${T}_other_configs.txt
rm -rf $O/stats $O/fetch $O/write $O/busy $O/c3
ls -la $O
