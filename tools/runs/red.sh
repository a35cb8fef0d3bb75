cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_rec_gpu.py tests/test_det_ops_gpu.py tests/test_det_model_gpu.py tests/test_train_loop_gpu.py -x -q -m gpu 2>&1 | tail -3
rm -rf gpurun_out/trace_crnn
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_crnn -- python tools/prof_crnn.py --steps 3 --warmup 2 > gpurun_out/trace_crnn.log 2>&1
python tools/trace_step.py gpurun_out/trace_crnn k_conv0_fwd > gpurun_out/crnn_step_trace.txt 2>&1
find gpurun_out/trace_crnn -name "*kernel_trace.csv" -delete
tail -2 gpurun_out/crnn_step_trace.txt
python bench.py --no-cpu-baseline --no-fp32 --no-gru-exact 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['crnn']['value'], d['crnn']['ms_per_step'])"
for v in "OCRS_REC_REDUCE_BPC=2" "OCRS_REC_REDUCE_BPC=8"; do env $v python tools/prof_crnn.py --steps 10 --warmup 3 2>&1 | tail -1 | cut -c1-200; done
