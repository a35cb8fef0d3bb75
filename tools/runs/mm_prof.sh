cd $GRAFT_REPO_ROOT
OCRS_LIB_PATH=$GRAFT_REPO_ROOT/ocrs_models_amd/variants/libocrs_hip_mmprof.so timeout 300 python tools/runs/mm_prof.py 2>&1 | grep -v amdgpu.ids | tail -9
