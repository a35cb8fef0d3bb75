cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trace_crnn
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_crnn -- python tools/prof_crnn.py --steps 3 --warmup 2 > gpurun_out/trace_crnn.log 2>&1
python - <<'PY'
import csv,glob,re
f=sorted(glob.glob('gpurun_out/trace_crnn/**/*kernel_stats.csv',recursive=True))[-1]
rows=list(csv.DictReader(open(f)))
out=open('gpurun_out/crnn_stats.txt','w')
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:40]:
    nm = re.sub(r'[(].*', '', r['Name'])[:70]
    out.write(f"{nm:70s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/5e3:10.1f} us/step {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']}\n")
out.write(f"total per step us: {tot/5e3}\n")
PY
find gpurun_out/trace_crnn -name "*kernel_trace.csv" -delete
tail -2 gpurun_out/trace_crnn.log
