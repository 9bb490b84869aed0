#!/bin/bash
# the hipGraph cache of the step engine under the conditions of round 5's open numerics error (tools/graph_probe.py:
# models built one after another in one process, FusedAdam and the verbatim loop, cache on / off), with the old
# eviction (16 graphs, destroyed at once) and the current one (32, retired behind an event); then tools/graph_stress.py
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3; do
  echo "### old eviction, run $i"
  GPSA_STEP_GRAPH_MAX=16 GPSA_STEP_GRAPH_UNSAFE_DESTROY=1 timeout 400 python tools/graph_probe.py 600 2>&1 | grep -v amdgpu.ids | grep -E "Error|error|cache on" | cut -c1-200 | tail -8
done
for i in 1 2 3; do
  echo "### current eviction, run $i"
  timeout 400 python tools/graph_probe.py 600 2>&1 | grep -v amdgpu.ids | grep -E "Error|error|cache on" | cut -c1-200 | tail -8
done
echo "### trajectories"; timeout 300 python tools/graph_stress.py 3 600 2>&1 | grep -v amdgpu.ids | tail -4
