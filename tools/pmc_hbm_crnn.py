"""HBM traffic AND achieved GB/s per kernel of the CRNN train step from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) plus the kernel
trace of the same runs -> profiles/<name>_crnn_pmc_hbm.csv  (north_star: "rocprof showing achieved HBM GB/s on the memory-bound
CTC / activation kernels").

usage: python tools/pmc_hbm_crnn.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <steps profiled> <out.csv>
Units / corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KB; on gfx950 FETCH_SIZE under-reports
wide coalesced reads by exactly 2x -> doubled; WRITE_SIZE as is.  Durations: kernel-trace timestamps of the FETCH pass (counter collection
serialises the kernels: per-kernel times are the un-overlapped ones, a few % above the free-running step).
"""
import collections
import csv
import glob
import re
import sys


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "")


def load(d, counter):
    acc = collections.defaultdict(float)
    n = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            acc[k] += float(r["Counter_Value"])
            n[k].add(r["Dispatch_Id"])
    return acc, {k: len(v) for k, v in n.items()}


def durations(d):
    us = collections.defaultdict(float)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            us[short(r["Kernel_Name"])] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
    return us


fd, wd, steps, out = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
fa, fn = load(fd, "FETCH_SIZE")
wa, wn = load(wd, "WRITE_SIZE")
us = durations(fd)
rows = []
for k in sorted(set(fa) | set(wa)):
    rd, wr = 2 * fa.get(k, 0.0) * 1024 / 1e9 / steps, wa.get(k, 0.0) * 1024 / 1e9 / steps
    t = us.get(k, 0.0) / steps
    rows.append([k, fn.get(k, wn.get(k, 0)) / steps, rd, wr, t, (rd + wr) * 1e9 / (t * 1e-6) / 1e9 if t > 0 else 0.0])
rows.sort(key=lambda r: -(r[2] + r[3]))
with open(out, "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["kernel", "launches_per_step", "fetch_GB_per_step_x2_corrected", "write_GB_per_step", "us_per_step", "achieved_GBps"])
    for r in rows:
        w.writerow([r[0], round(r[1], 3), round(r[2], 5), round(r[3], 5), round(r[4], 2), round(r[5], 1)])
    w.writerow(["TOTAL", "", round(sum(r[2] for r in rows), 4), round(sum(r[3] for r in rows), 4), round(sum(r[4] for r in rows), 1), ""])
print("CRNN step: HBM read", round(sum(r[2] for r in rows), 3), "GB, write", round(sum(r[3] for r in rows), 3), "GB,", round(sum(r[4] for r in rows), 1), "us of kernels")
for r in rows[:25]:
    print(f"  {r[0][:60]:60s} {r[1]:5.1f} launches {1e3 * (r[2] + r[3]):9.2f} MB {r[4]:8.1f} us {r[5]:8.1f} GB/s")
