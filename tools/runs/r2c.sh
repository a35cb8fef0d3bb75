cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
python -m pytest tests/test_det_ops_gpu.py -m gpu -q -s -k "matrix_core" > gpurun_out/r2c/ops.log 2>&1
tail -5 gpurun_out/r2c/ops.log
python -m pytest tests/test_det_ops_gpu.py tests/test_det_model_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -k "not matrix_core" > gpurun_out/r2c/model.log 2>&1
tail -5 gpurun_out/r2c/model.log
python -m pytest tests/test_full_size_gpu.py tests/test_train_loop_gpu.py -m gpu -q -s > gpurun_out/r2c/new_tests.log 2>&1
tail -15 gpurun_out/r2c/new_tests.log
bash tools/runs/prof_stats.sh r2c_mm
