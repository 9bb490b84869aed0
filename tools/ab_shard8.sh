#!/bin/bash
# A/B of rank 0's share of an 8-way row sharding on one GPU (bench.py --emulate-shard 8): KL terms at weight 1/world on
# every rank ("scaled", rounds 1-5) against owner-computes ("owner", round 6); eager and one-graph replay, interleaved.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2 3; do
  for m in scaled owner; do
    python3 bench.py --no-cpu-baseline --no-extras --blocks 5 --emulate-shard ${1:-8} --kl-share $m 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
g=d.get('graph_replay') or {}
print('$m', 'eager ms', round(d['ms_per_step'],4), 'S1', round(d['secondary_S1']['ms_per_step'],4), 'graph ms', round(g.get('ms_per_step', float('nan')),4))"
  done
done
