cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
python -m pytest tests/test_det_ops_gpu.py -m gpu -q -s -k "matrix_core or dwpw_block or maxpool or fused_bn" > gpurun_out/r2b/ops.log 2>&1
tail -5 gpurun_out/r2b/ops.log
python -m pytest tests/test_det_model_gpu.py tests/test_edge_cases_gpu.py tests/test_input_pipeline_gpu.py -m gpu -q > gpurun_out/r2b/model.log 2>&1
tail -5 gpurun_out/r2b/model.log
python -m pytest tests/test_full_size_gpu.py tests/test_train_loop_gpu.py -m gpu -q -s > gpurun_out/r2b/new_tests.log 2>&1
tail -15 gpurun_out/r2b/new_tests.log
timeout 600 python bench.py --no-cpu-baseline --no-crnn --no-fp32 > gpurun_out/r2b/bench_mm.json 2> gpurun_out/r2b/bench_mm.err
cat gpurun_out/r2b/bench_mm.json; tail -3 gpurun_out/r2b/bench_mm.err
OCRS_MM=0 timeout 600 python bench.py --no-cpu-baseline --no-crnn --no-fp32 > gpurun_out/r2b/bench_old.json 2> gpurun_out/r2b/bench_old.err
cat gpurun_out/r2b/bench_old.json
timeout 300 python bench.py --no-roofline --no-crnn --no-cpu-baseline --no-fp32 > gpurun_out/r2b/bench_noroof.json 2>&1
cat gpurun_out/r2b/bench_noroof.json
