cd $GRAFT_REPO_ROOT
timeout 300 python tools/runs/gru_prof2.py 2>&1 | grep -v amdgpu.ids | tail
