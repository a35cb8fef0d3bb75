"""Per-launch durations of ONE detection train step out of a rocprofv3 --kernel-trace CSV.

usage: python tools/trace_step.py <dir with *_kernel_trace.csv> [marker-kernel]   (marker default: k_bce_fwd = one per step)
Prints launch order, short kernel name, grid, duration (us) for the last complete step, then per-name totals.
"""
import csv, glob, re, sys, collections

d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "k_bce_fwd"
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
lo, hi = idx[-2], idx[-1]
tot = collections.OrderedDict()
t0 = int(rows[lo]["Start_Timestamp"])
for i in range(lo, hi):
    r = rows[i]
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{i-lo:4d} {(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {name:42s} grid={r.get('Grid_Size_X', r.get('Grid_Size','?')):>8s} {us:9.1f}")
    tot[name] = tot.get(name, 0) + us
print("---- totals (us)")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"{k:42s} {v:9.1f}")
print("step span us:", (int(rows[hi]["Start_Timestamp"]) - t0) / 1e3, " kernel sum:", sum(tot.values()))
