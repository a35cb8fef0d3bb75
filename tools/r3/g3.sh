cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_det_ops_gpu.py -q -x --tb=short -k "matrix_core_block_forward" 2>&1 | tail -15
echo "=== DMA on"; timeout 300 python tools/r3/mm_time.py --fwd 2>&1 | grep -v amdgpu.ids | grep fwd
echo "=== DMA off"; OCRS_MM_DMA=0 timeout 300 python tools/r3/mm_time.py --fwd 2>&1 | grep -v amdgpu.ids | grep fwd
