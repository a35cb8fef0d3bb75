cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/trace_crnn
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_crnn -- python tools/prof_crnn.py --steps 3 --warmup 2 > gpurun_out/trace_crnn.log 2>&1
python tools/trace_step.py gpurun_out/trace_crnn k_conv0_fwd > gpurun_out/crnn_step_trace.txt 2>&1
find gpurun_out/trace_crnn -name "*kernel_trace.csv" -delete
tail -2 gpurun_out/crnn_step_trace.txt
python tools/prof_crnn.py --steps 10 --warmup 3 2>&1 | tail -1 | cut -c1-160
