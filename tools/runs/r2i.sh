cd $GRAFT_REPO_ROOT
python -m pytest tests/test_det_ops_gpu.py -x -q -k "matrix_core" 2>&1 | tail -2
bash tools/runs/variants.sh "$@" 2>&1 | grep -E "==|k_mm_fwd<"
