cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in default ths4; do
  rm -rf gpurun_out/kp
  if [ "$v" = default ]; then E=""; else E="OCRS_LIB_PATH=$GRAFT_REPO_ROOT/ocrs_models_amd/variants/libocrs_hip_$v.so"; fi
  env $E timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kp -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-crnn --no-fp32 --no-roofline > gpurun_out/kp.json 2>/dev/null
  f=$(find gpurun_out/kp -name "*kernel_stats.csv" | head -1)
  echo "== $v $(head -c 150 gpurun_out/kp.json | grep -o '"ms_per_step": [0-9.]*')"
  python -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'k_pwb<64' in r['Name']: print('  ', r['Name'][:44], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us avg')
"
done
rm -rf gpurun_out/kp
