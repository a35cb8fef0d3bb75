"""HBM traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) -> profiles/<name>_pmc_hbm.csv.

usage: python tools/pmc_hbm.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <steps profiled> <out.csv>
Units / corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KB; on gfx950 FETCH_SIZE
under-reports wide coalesced reads by exactly 2x -> the corrected column doubles it; WRITE_SIZE is taken as is.
"""
import csv, glob, re, sys, collections

def load(d, counter):
    acc = collections.defaultdict(float); n = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter: continue
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            acc[name] += float(r["Counter_Value"]); n[name].add(r["Dispatch_Id"])
    return acc, {k: len(v) for k, v in n.items()}

fd, wd, steps, out = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
fa, fn = load(fd, "FETCH_SIZE"); wa, wn = load(wd, "WRITE_SIZE")
rows = []
for k in sorted(set(fa) | set(wa)):
    rows.append([k, fn.get(k, wn.get(k, 0)) / steps, 2 * fa.get(k, 0.0) * 1024 / 1e9 / steps, wa.get(k, 0.0) * 1024 / 1e9 / steps])
rows.sort(key=lambda r: -(r[2] + r[3]))
with open(out, "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["kernel", "launches_per_step", "fetch_GB_per_step_x2_corrected", "write_GB_per_step"])
    for r in rows: w.writerow([r[0], round(r[1], 3), round(r[2], 4), round(r[3], 4)])
    w.writerow(["TOTAL", "", round(sum(r[2] for r in rows), 3), round(sum(r[3] for r in rows), 3)])
print("total GB/step: read", round(sum(r[2] for r in rows), 2), "write", round(sum(r[3] for r in rows), 2))
