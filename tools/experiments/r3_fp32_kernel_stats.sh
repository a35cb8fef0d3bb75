#!/bin/bash
# per-kernel time of the fp32 parity-mode detection step (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ks32
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks32 -- python bench.py --dtype fp32 --steps 3 --warmup 1 --no-crnn --no-cpu-baseline --no-roofline --no-fp32 --no-ref-style --no-ddp-probe --no-config1 > gpurun_out/ks32.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/ks32/**/*kernel_stats.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms over 4 steps", tot / 1e6)
for r in rows[:30]:
    print(f'{float(r["TotalDurationNs"]) / 1e6:8.2f} ms {int(r["Calls"]):5d} {float(r["AverageNs"]) / 1e3:8.1f} us {float(r["TotalDurationNs"]) / tot * 100:5.1f}%  {r["Name"][:90]}')
PY
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/ks32/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for pat in ("k_bn_bwd_reduce", "k_wgrad_gather<", "k_dw_bwd<float, 2", "k_pw_bwd<float, 8, 8>", "k_dwpw_fwd<float, 1, 1"):
    sel = [r for r in rows if pat in r["Kernel_Name"]]
    n = len(sel) // 4
    print(pat, [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in sel[-n:]], [r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size") for r in sel[-n:]])
PY
find gpurun_out/ks32 -name "*.csv" -size +1M -delete
