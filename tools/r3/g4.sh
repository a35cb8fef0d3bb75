cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 600 python -m pytest tests/test_rec_gpu.py tests/test_gru_gpu.py -q -x --tb=short -k "timeout or eval_mode or golden or gru" 2>&1 | tail -5
( time timeout 900 python bench.py > gpurun_out/r3/bench1.json 2> gpurun_out/r3/bench1.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench1.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')})
print('ddp', d.get('ddp'))
print('ref_style', d.get('reference_style_step'))
print('cpu', {k:v for k,v in d.get('cpu_baseline',{}).items() if k in ('value','cores','det_B4_1024','all_physical_cores')})
print('crnn', d['crnn']['value'], d['crnn']['ms_per_step'])
PY
tail -3 gpurun_out/r3/bench1.err
bash tools/run_pmc_hbm_crnn.sh r03_crnn_pmc_hbm.csv 2>&1 | tail -30
