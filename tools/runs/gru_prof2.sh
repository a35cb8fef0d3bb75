cd $GRAFT_REPO_ROOT
OCRS_LIB_PATH=$GRAFT_REPO_ROOT/ocrs_models_amd/variants/libocrs_hip_gruprof.so timeout 300 python tools/runs/gru_prof2.py 2>&1 | grep -v amdgpu.ids | tail
