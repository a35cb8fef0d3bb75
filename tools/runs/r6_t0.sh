cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r6_t0_tests.log 2>&1; tail -3 gpurun_out/r6_t0_tests.log
rm -rf gpurun_out/trace32
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace32 -- python bench.py --dtype fp32 --steps 3 --warmup 2 --no-crnn --no-cpu-baseline --no-roofline --no-fp32 --no-ref-style --no-ddp-probe --no-config1 --no-pmc > gpurun_out/trace32.log 2>&1
python tools/trace_step.py gpurun_out/trace32 > gpurun_out/r6_fp32_step_trace.txt 2>&1
find gpurun_out/trace32 -name "*kernel_trace.csv" -delete
tail -4 gpurun_out/r6_fp32_step_trace.txt; tail -2 gpurun_out/trace32.log | cut -c1-600
