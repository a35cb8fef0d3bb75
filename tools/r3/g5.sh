cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_det_ops_gpu.py -q -x --tb=short -k "matrix_core_block_forward" 2>&1 | tail -4
echo "=== FULL on"; timeout 300 python tools/r3/mm_time.py --fwd 2>&1 | grep -v amdgpu.ids | grep fwd
echo "=== FULL off"; OCRS_MM_FULL=0 timeout 300 python tools/r3/mm_time.py --fwd 2>&1 | grep -v amdgpu.ids | grep fwd
