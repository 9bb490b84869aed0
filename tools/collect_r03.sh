#!/bin/bash
# Round-3 numbers for the docs: other configurations, strong-scaling emulation, the L-axis bench path on one device.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
for w in 3 4cut 5cut; do python3 tools/run_config.py $w 6 3 2>/dev/null | tail -4; done > $O/other_configs.txt
python3 tools/run_config.py 4 3 2 2>/dev/null | tail -4 >> $O/other_configs.txt
python3 tools/run_config.py 5 2 1 2>/dev/null | tail -4 >> $O/other_configs.txt
GPSA_KEEP_GB=0 python3 tools/run_config.py 4 3 2 2>/dev/null | tail -4 | sed 's/^/[GPSA_KEEP_GB=0: nothing kept] /' >> $O/other_configs.txt
for k in 1 2 4 8; do python3 bench.py --no-cpu-baseline --emulate-shard $k 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('shard 1/$k  ms_per_step', round(d['ms_per_step'],3), ' graph', round(d['graph_replay']['ms_per_step'],3))"; done > $O/strong_scaling_emulation.txt
GPSA_BENCH_ONE_DEVICE=1 python3 bench.py --gpus 2 --workload 4 --shard outputs --steps 2 --warmup 1 > $O/bench_c4_outputs_2ranks_one_device.json 2> $O/bench_c4_outputs.err
python3 bench.py --workload 4 --steps 3 --warmup 2 > $O/bench_c4_1gpu.json 2>/dev/null
cat $O/other_configs.txt $O/strong_scaling_emulation.txt
head -c 400 $O/bench_c4_outputs_2ranks_one_device.json; echo
tail -3 $O/bench_c4_outputs.err
