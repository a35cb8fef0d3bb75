#!/usr/bin/env python3
"""Print VGPR / scratch / occupancy / LDS per kernel of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage)."""
import os, re, subprocess, sys
src = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC", "-c", src, "-o", "/tmp/kres.o", *os.environ.get("KRES_FLAGS", "").split(),
                    "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.rsplit(":", 1); rows[cur][k.strip()] = v.strip()
if r.returncode != 0 or not rows:
    sys.stderr.write(r.stderr[-3000:]); sys.exit("compile failed")
dem = subprocess.run(["c++filt"] + list(rows), capture_output=True, text=True, stdin=subprocess.DEVNULL).stdout.splitlines()
for name, d in zip(dem, rows.values()):
    if pat and not re.search(pat, name): continue
    short = re.sub(r"\(.*", "", name).replace("void ", "")
    print(f"{short:45s} vgpr {d.get('VGPRs','?'):>4s} agpr {d.get('AGPRs','?'):>3s} scratch {d.get('ScratchSize [bytes/lane]','?'):>4s} occ {d.get('Occupancy [waves/SIMD]','?')} lds {d.get('LDS Size [bytes/block]','?')}")
