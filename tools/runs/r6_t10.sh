cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_det_ops_gpu.py -x -q -k "rs32 or conv_transpose or dwpw_block or fused_bn or maxpool" > gpurun_out/r6_t10_ops.log 2>&1; tail -2 gpurun_out/r6_t10_ops.log
timeout 900 python -m pytest tests/test_det_model_gpu.py tests/test_edge_cases_gpu.py tests/test_full_size_gpu.py tests/test_train_loop_gpu.py -x -q > gpurun_out/r6_t10_model.log 2>&1; tail -2 gpurun_out/r6_t10_model.log
python bench.py --steps 10 --warmup 3 --no-crnn --no-cpu-baseline --no-ref-style --no-ddp-probe --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['fp32_exact']['value'], d['fp32_exact']['ms_per_step'], d['config1_hip'])"
