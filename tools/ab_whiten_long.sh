cd $GRAFT_REPO_ROOT
export GPSA_TW_SHORT=2
for c in 0 1; do
for s in 0 15; do
  echo "## CONST=$c SKIP=$s"
  GPSA_TW_CONST=$c GPSA_PROJ64=1 GPSA_PROJ64_SKIP=$s python tools/time_whiten.py 10 2>&1 | grep -v amdgpu.ids | grep "C = 400"
done
done
