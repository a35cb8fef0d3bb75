cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_det_ops_gpu.py tests/test_det_model_gpu.py tests/test_edge_cases_gpu.py -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for o in 1 0; do
  rm -rf gpurun_out/kp
  OCRS_CTD=$o timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kp -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-crnn --no-fp32 --no-roofline > gpurun_out/kp.json 2>/dev/null
  f=$(find gpurun_out/kp -name "*kernel_stats.csv" | head -1)
  echo "== OCRS_CTD=$o $(head -c 150 gpurun_out/kp.json | grep -o '"ms_per_step": [0-9.]*')"
  python -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'k_ctd' in r['Name'] or 'k_convt_dgrad<' in r['Name']: print('  ', r['Name'][:44], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us avg', round(float(r['TotalDurationNs'])/6e3,1), 'us/step')
"
done
rm -rf gpurun_out/kp
