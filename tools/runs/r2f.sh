# pooled-quad experiment: parity tests of the matrix-core kernels, then per-kernel stats of the variants
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_det_ops_gpu.py -x -q -k "matrix_core" 2>&1 | tail -5
bash tools/runs/variants.sh default p8 2>&1 | grep -E "==|true, true, true|32, 32, true"
