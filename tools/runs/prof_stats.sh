# usage: bash tools/runs/prof_stats.sh <tag> [env assignments...]   -> gpurun_out/<tag>_kstats.txt (top kernels, per-step ms)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ks_$TAG
env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$TAG -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-crnn --no-fp32 --no-roofline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_err.log
f=$(find gpurun_out/ks_$TAG -name "*kernel_stats.csv" | head -1)
python - "$f" 6 <<'PY' > gpurun_out/${TAG}_kstats.txt
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
tot = 0
out = []
for r in rows:
    name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
    ms = float(r["TotalDurationNs"]) / 1e6 / steps
    out.append((ms, int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, name))
    tot += ms
out.sort(reverse=True)
print(f"total kernel ms/step {tot:.3f}")
for ms, calls, avg, name in out[:45]:
    print(f"{ms:8.3f} ms/step  {calls:6.1f} calls/step  {avg:9.1f} us avg  {name[:90]}")
PY
cp $f gpurun_out/${TAG}_kernel_stats.csv
rm -rf gpurun_out/ks_$TAG
cat gpurun_out/${TAG}_bench.json | head -c 400; echo
head -40 gpurun_out/${TAG}_kstats.txt
