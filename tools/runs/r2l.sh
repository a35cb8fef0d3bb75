cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_det_ops_gpu.py tests/test_det_model_gpu.py tests/test_edge_cases_gpu.py -x -q 2>&1 | tail -3
for i in 1 2; do python bench.py --no-crnn --no-cpu-baseline --no-fp32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['ms_per_step'])"; done
