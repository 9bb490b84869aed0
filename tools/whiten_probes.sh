#!/bin/bash
# The probes behind LAB_NOTES' "projection kernel" section, on one box (tools/time_whiten.py, C = 400 000 panels):
#   skip   timing-only variants (GPSA_PROJ64_SKIP: 1 stores, 2 column re-loads, 4 LDS-DMA, 8 waits)
#   occ    one / two / three workgroups per CU, full and stripped
#   const  constant against random operands, full and stripped
#   pitch  one long panel against many short ones (row pitch), new and old kernel
# usage (GPU box, repo root): bash tools/whiten_probes.sh skip|occ|const|pitch
cd ${GRAFT_REPO_ROOT:-.}
T="python tools/time_whiten.py 10"
f() { grep -v amdgpu.ids | grep -E "C = 400|columns|^##"; }
case ${1:-skip} in
  skip)
    export GPSA_TW_SHORT=2
    for s in 0 1 3 7 15; do echo "## GPSA_PROJ64_SKIP=$s"; GPSA_PROJ64_SKIP=$s $T 2>&1 | f; done
    echo "## whiten_mfma_kernel"; GPSA_PROJ64=0 $T 2>&1 | f ;;
  occ)
    export GPSA_TW_SHORT=2
    for o in 1 2 3; do for s in 0 15; do echo "## GPSA_PROJ64_OCC=$o GPSA_PROJ64_SKIP=$s"; GPSA_PROJ64_OCC=$o GPSA_PROJ64_SKIP=$s $T 2>&1 | f; done; done ;;
  const)
    export GPSA_TW_SHORT=2
    for c in 0 1; do for s in 0 15; do echo "## GPSA_TW_CONST=$c GPSA_PROJ64_SKIP=$s"; GPSA_TW_CONST=$c GPSA_PROJ64_SKIP=$s $T 2>&1 | f; done; done ;;
  pitch)
    export GPSA_TW_SHORT=3
    echo "## proj64_kernel"; GPSA_PROJ64_MIN_TILES=0 $T 2>&1 | f
    echo "## whiten_mfma_kernel"; GPSA_PROJ64=0 $T 2>&1 | f ;;
esac
