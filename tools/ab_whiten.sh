#!/bin/bash
# whiten_mfma_kernel (GPSA_PROJ64=0) against proj64_kernel at the step's shapes, one box, tests first
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_batched.py -q -m gpu -x -k "whiten" 2>&1 | tail -3
{
  GPSA_PROJ64=0 python tools/time_whiten.py 20
  GPSA_PROJ64=1 GPSA_PROJ64_MIN_TILES=0 python tools/time_whiten.py 20
  GPSA_PROJ64=1 GPSA_PROJ64_MIN_TILES=0 GPSA_PROJ64_OCC=2 python tools/time_whiten.py 20
} 2>&1 | grep -v amdgpu.ids > $O/r05_whiten_ab.txt
cat $O/r05_whiten_ab.txt
