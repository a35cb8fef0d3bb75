cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_det_ops_gpu.py tests/test_rec_gpu.py tests/test_det_model_gpu.py tests/test_edge_cases_gpu.py tests/test_full_size_gpu.py -x -q -m gpu 2>&1 | tail -3
bash tools/run_trace_step.sh > /dev/null 2>&1; grep -n "k_wgrad_gather\|k_pwb<128, 256, true\|step span" gpurun_out/trace_step.txt | head -12
python bench.py --no-cpu-baseline --no-fp32 --no-gru-exact 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['passes']['block_bwd']['ms_per_step'], d['roofline']['passes']['convt_bwd'], d['crnn']['value'], d['crnn']['ms_per_step'])"
