cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_det_ops_gpu.py -x -q -k "conv_transpose" > gpurun_out/r6_t6_ops.log 2>&1; tail -4 gpurun_out/r6_t6_ops.log
timeout 900 python -m pytest tests/test_det_model_gpu.py tests/test_edge_cases_gpu.py tests/test_full_size_gpu.py -x -q > gpurun_out/r6_t6_model.log 2>&1; tail -4 gpurun_out/r6_t6_model.log
rm -rf gpurun_out/trace32
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace32 -- python bench.py --dtype fp32 --steps 3 --warmup 2 --no-crnn --no-cpu-baseline --no-roofline --no-fp32 --no-ref-style --no-ddp-probe --no-config1 --no-pmc > gpurun_out/trace32.log 2>&1
python tools/trace_step.py gpurun_out/trace32 > gpurun_out/r6_fp32_step_trace_t6.txt 2>&1
find gpurun_out/trace32 -name "*kernel_trace.csv" -delete
grep "k_rs32_ct\|k_convt\|step span" gpurun_out/r6_fp32_step_trace_t6.txt | head -40; tail -1 gpurun_out/trace32.log | cut -c1-300
