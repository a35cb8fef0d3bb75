#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ctc_tr
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ctc_tr -- python tools/experiments/r4_ctc_check.py > gpurun_out/ctc_tr.log 2>&1
python - <<'PY'
import csv,glob,re,collections
f=sorted(glob.glob('gpurun_out/ctc_tr/**/*kernel_trace.csv',recursive=True))[-1]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    nm=re.sub(r'[(].*','',r['Kernel_Name']).replace('void ','')
    if 'ctc' in nm: d[(nm,r['Grid_Size_X'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items()):
    v=sorted(v); print(f"{k[0][:44]:44s} grid {k[1]:>7s} n={len(v):3d} median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f}")
PY
find gpurun_out/ctc_tr -name "*.csv" -size +1M -delete
