cd $GRAFT_REPO_ROOT
python tools/power_probe.py whiten_bwd 3 2>&1 | grep -v amdgpu.ids | grep -E "launches|busy|whiten" | sed 's/=\{5,\}[^|]*|//g' | cut -c1-220
GPSA_TW_CONST=1 python tools/power_probe.py whiten_bwd 3 2>&1 | grep -v amdgpu.ids | grep -E "launches|busy|whiten" | cut -c1-220
GPSA_TW_CONST=1 python tools/power_probe.py whiten_fwd 3 2>&1 | grep -v amdgpu.ids | grep -E "launches|busy|whiten" | cut -c1-220
GPSA_PROJ64=0 python tools/power_probe.py whiten_bwd 3 2>&1 | grep -v amdgpu.ids | grep -E "launches|busy|whiten" | cut -c1-220
GPSA_PROJ64=0 python tools/power_probe.py whiten_fwd 3 2>&1 | grep -v amdgpu.ids | grep -E "launches|busy|whiten" | cut -c1-220
