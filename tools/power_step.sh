#!/bin/bash
# package power while the headline step runs back to back (rocm-smi sampled from the shell every ~0.5 s)
cd ${GRAFT_REPO_ROOT:-.}
python bench.py --headline-only --no-traffic --blocks 1 --steps ${1:-1500} --warmup 5 > /tmp/ps_bench.log 2>&1 &
BP=$!
for i in $(seq 1 60); do
  sleep 0.5
  if ! kill -0 $BP 2>/dev/null; then break; fi
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/GPU\[0\]\t*: //' | tr '\n' ' '
  echo
done | awk '{print}' | grep -v "Power (W): 2[0-9][0-9]\." | tail -12
wait $BP
python3 -c "import json; d=json.loads(open('/tmp/ps_bench.log').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"
