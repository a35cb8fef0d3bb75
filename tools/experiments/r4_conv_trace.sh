#!/bin/bash
# kernel-only durations (rocprofv3 kernel trace) of the conv kernels in tools/experiments/r4_conv_time.py
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ctr
R4_SHAPES=${R4_SHAPES:-3} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ctr -- python tools/experiments/r4_conv_time.py > gpurun_out/ctr.log 2>&1
python - <<'PY'
import csv,glob,re,collections
f=sorted(glob.glob('gpurun_out/ctr/**/*kernel_trace.csv',recursive=True))[-1]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    nm=re.sub(r'[(].*','',r['Kernel_Name']).replace('void ','')
    if 'conv3x3' in nm or 'pack' in nm: d[(nm,r['Grid_Size'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items():
    v=sorted(v); print(f"{k[0][:40]:40s} grid {k[1]:>8s} n={len(v):3d} median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f}")
PY
grep N=256 gpurun_out/ctr.log | cut -c1-100
find gpurun_out/ctr -name "*.csv" -size +1M -delete
