cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -q -x --deselect tests/test_full_size_gpu.py --deselect tests/test_train_loop_gpu.py > gpurun_out/r2a/old_tests.log 2>&1
tail -3 gpurun_out/r2a/old_tests.log
python -m pytest tests/test_full_size_gpu.py tests/test_train_loop_gpu.py -m gpu -q -s > gpurun_out/r2a/new_tests.log 2>&1
tail -40 gpurun_out/r2a/new_tests.log
timeout 900 python bench.py --rec-config5 > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
tail -c 5000 gpurun_out/r2a/bench.json; tail -5 gpurun_out/r2a/bench.err
timeout 300 python bench.py --no-roofline --no-crnn --no-cpu-baseline --no-fp32 > gpurun_out/r2a/bench_noroof.json 2>&1
cat gpurun_out/r2a/bench_noroof.json
