cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python bench.py --steps 3 --warmup 2 --no-crnn --no-cpu-baseline --no-roofline --no-fp32 --no-ref-style --no-ddp-probe --no-config1 > gpurun_out/trace_bench.log 2>&1
python tools/trace_step.py gpurun_out/trace > gpurun_out/trace_step.txt 2>&1
find gpurun_out/trace -name "*kernel_trace.csv" -delete
tail -5 gpurun_out/trace_step.txt
