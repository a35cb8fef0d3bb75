"""Instruction mix per basic block of one kernel in a hipcc -S dump.  usage: isa_mix.py file.s <mangled-substring>"""
import re, sys, collections
txt = open(sys.argv[1]).read().splitlines()
key = sys.argv[2]
start = next(i for i, l in enumerate(txt) if l.startswith("_Z") and key in l.split(":")[0])
end = next(i for i in range(start, len(txt)) if txt[i].strip().startswith("s_endpgm"))
blocks = []; cur = ("entry", collections.Counter(), [])
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_pk_"): return "vpk"
    if op.startswith("v_cvt"): return "vcvt"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return op
    if op.startswith(("global_", "buffer_", "scratch_")): return op
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    return "other"
for l in txt[start + 1:end + 1]:
    s = l.strip()
    m = re.match(r"(\.LBB\w+):", s)
    if m:
        blocks.append(cur); cur = (m.group(1), collections.Counter(), []); continue
    m = re.match(r"([a-z][a-z_0-9]+)\b(.*)", s)
    if not m or s.startswith((";", ".")): continue
    cur[1][cls(m.group(1))] += 1
    if m.group(1).startswith(("s_cbranch", "s_branch")): cur[2].append(m.group(2).strip())
blocks.append(cur)
names = [b[0] for b in blocks]
tot = collections.Counter()
for i, (n, c, br) in enumerate(blocks):
    back = [t for t in br if t in names and names.index(t) <= i]
    tot.update(c)
    if sum(c.values()) >= 8 or back:
        print(f"{n:12s} n={sum(c.values()):4d} {'BACK->' + ','.join(back) if back else '':18s}", dict(sorted(c.items())))
print("TOTAL", dict(sorted(tot.items())))
