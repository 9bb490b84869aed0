cd $GRAFT_REPO_ROOT
export GPSA_TW_SHORT=3
GPSA_PROJ64=1 GPSA_PROJ64_MIN_TILES=0 python tools/time_whiten.py 10 2>&1 | grep -v amdgpu.ids
GPSA_PROJ64=0 python tools/time_whiten.py 10 2>&1 | grep -v amdgpu.ids
