#!/bin/bash
# usage: tools/kres2.sh <file.hip> [extra flags]: compile one source for gfx950 and print per-kernel SGPR/VGPR/AGPR/scratch/occupancy
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -c $f -o /tmp/kres2.o -Rpass-analysis=kernel-resource-usage "$@" 2> /tmp/kres2.log
grep -E "error" /tmp/kres2.log | head -20
python3 - <<'PY'
import re, subprocess
t=open('/tmp/kres2.log').read()
for m in re.finditer(r"Function Name: (\S+).*?SGPRs: (\d+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)", t, re.S):
    name=subprocess.run(['c++filt',m.group(1)],capture_output=True,text=True).stdout.strip()
    name=re.sub(r"\(.*","",name).replace("void ","")
    print(name[:64].ljust(64), 'sgpr',m.group(2).rjust(3),'vgpr',m.group(3).rjust(3),'agpr',m.group(4).rjust(3),'scratch',m.group(5).rjust(4),'occ',m.group(6))
PY
