cd $GRAFT_REPO_ROOT
OCRS_LIB_PATH=$GRAFT_REPO_ROOT/ocrs_models_amd/variants/libocrs_hip_pfd2.so timeout 300 python tools/r3/mm_time.py 2>&1 | grep -v amdgpu.ids | tail -20
