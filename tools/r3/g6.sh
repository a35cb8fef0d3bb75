cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_det_ops_gpu.py -q -x --tb=short -k "matrix_core_block_forward or rounding_matched" 2>&1 | tail -4
echo "=== FULL on"; timeout 300 python tools/r3/mm_time.py 2>&1 | grep -v amdgpu.ids | grep bwd
echo "=== FULL off"; OCRS_MM_FULL=0 timeout 300 python tools/r3/mm_time.py 2>&1 | grep -v amdgpu.ids | grep bwd
