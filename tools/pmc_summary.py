"""Per-kernel mean of rocprofv3 --pmc counters.  usage: pmc_summary.py <dir> [name-regex]"""
import csv, glob, re, sys, collections
d = sys.argv[1]; pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if pat and not pat.search(name): continue
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"]); n[name].add((f, r["Dispatch_Id"]))
for name in sorted(acc, key=lambda k: -len(n[k])):
    cnt = len(n[name])
    print(f"{name}  dispatches={cnt}")
    for c, v in sorted(acc[name].items()): print(f"    {c:28s} {v / cnt:16.1f}")
