cd $GRAFT_REPO_ROOT
echo "=== PFD=2 variant"; OCRS_LIB_PATH=$GRAFT_REPO_ROOT/ocrs_models_amd/variants/libocrs_hip_pfd2.so timeout 300 python tools/r3/mm_time.py 2>&1 | grep -v amdgpu.ids | grep bwd
echo "=== PFD=1 (tree)"; timeout 300 python tools/r3/mm_time.py 2>&1 | grep -v amdgpu.ids | grep bwd
OCRS_LIB_PATH=$GRAFT_REPO_ROOT/ocrs_models_amd/variants/libocrs_hip_pfd2.so timeout 900 python -m pytest tests/test_det_ops_gpu.py -q -x --tb=short -k "rounding_matched" 2>&1 | tail -3
