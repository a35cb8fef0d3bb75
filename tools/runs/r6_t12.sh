cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_det_ops_gpu.py -x -q -k "rs32 or head" > gpurun_out/r6_t12_ops.log 2>&1; tail -4 gpurun_out/r6_t12_ops.log
timeout 900 python -m pytest tests/test_det_model_gpu.py tests/test_edge_cases_gpu.py tests/test_full_size_gpu.py tests/test_train_loop_gpu.py -x -q > gpurun_out/r6_t12_model.log 2>&1; tail -3 gpurun_out/r6_t12_model.log
B="python bench.py --dtype fp32 --steps 6 --warmup 2 --no-crnn --no-cpu-baseline --no-roofline --no-fp32 --no-ref-style --no-ddp-probe --no-config1 --no-pmc"
run() { echo "== $*"; env "$@" $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
run OCRS_RS32_HEAD=1
run OCRS_RS32_HEAD=0
run OCRS_RS32_HEAD=1
rm -rf gpurun_out/trace32
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace32 -- $B > gpurun_out/trace32.log 2>&1
python tools/trace_step.py gpurun_out/trace32 > gpurun_out/r6_fp32_step_trace_t12.txt 2>&1
find gpurun_out/trace32 -name "*kernel_trace.csv" -delete
head -14 gpurun_out/r6_fp32_step_trace_t12.txt | cut -c1-110; grep "step span" gpurun_out/r6_fp32_step_trace_t12.txt
