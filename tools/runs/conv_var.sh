cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  rm -rf gpurun_out/cv
  if [ "$v" = default ]; then E=""; else E="OCRS_LIB_PATH=$GRAFT_REPO_ROOT/ocrs_models_amd/variants/libocrs_hip_$v.so"; fi
  env $E timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cv -- python tools/runs/gru_dbg.py > gpurun_out/cv.log 2>&1
  f=$(find gpurun_out/cv -name "*kernel_stats.csv" | head -1)
  echo "== $v"; python -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'c128' in r['Name'] or 'conv_igemm<bf16' in r['Name'] or 'wgrad_tr' in r['Name']: print('  ', r['Name'][:40], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us avg')
"
done
rm -rf gpurun_out/cv
