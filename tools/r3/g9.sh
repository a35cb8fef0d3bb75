cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 2400 python -m pytest tests/test_det_ops_gpu.py tests/test_det_model_gpu.py tests/test_det_bf16_layerwise_gpu.py tests/test_edge_cases_gpu.py tests/test_full_size_gpu.py -q -x --tb=short 2>&1 | tail -6
timeout 600 python bench.py --no-cpu-baseline --no-crnn --no-fp32 --no-ref-style --no-ddp-probe > gpurun_out/r3/bench2.json 2> gpurun_out/r3/bench2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench2.json').read().strip().splitlines()[0])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['passes'].items()})
PY
OCRS_MM_FULL=0 timeout 600 python bench.py --no-cpu-baseline --no-crnn --no-fp32 --no-ref-style --no-ddp-probe > gpurun_out/r3/bench3.json 2> gpurun_out/r3/bench3.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench3.json').read().strip().splitlines()[0])
print("FULL=0", {k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['passes'].items()})
PY
