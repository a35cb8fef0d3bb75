cd $GRAFT_REPO_ROOT
timeout 300 python tools/runs/gru_dbg.py 2>&1 | grep -v amdgpu.ids | tail -12
