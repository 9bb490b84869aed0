#!/bin/bash
# where proj64_kernel's time goes: timing-only variants (GPSA_PROJ64_SKIP bits: 1 stores, 2 slab reloads, 4 LDS-DMA, 8 waits)
# x workgroups per CU (GPSA_PROJ64_OCC)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
export GPSA_TW_SHORT=1
for occ in 3 2; do
for s in ${1:-0 1 3 7 15}; do
  echo "## GPSA_PROJ64_OCC=$occ"
  GPSA_PROJ64=1 GPSA_PROJ64_OCC=$occ GPSA_PROJ64_SKIP=$s python tools/time_whiten.py 20 2>&1 | grep -v amdgpu.ids
done
done > $O/r05_whiten_skip.txt 2>&1
cat $O/r05_whiten_skip.txt
