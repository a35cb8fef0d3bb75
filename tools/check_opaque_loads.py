"""Build-time guard for the hand-waited prefetch loads of csrc/det_mm.hip (gload16_opaque + wait_vm_tied).

hipcc believes an asm load has completed when the statement ends, so if its register allocator spills, copies or re-uses the destination
vector before the hand-written `s_waitcnt` the kernel reads or stores stale bytes.  This script disassembles det_mm.hip (hipcc -S, device
only) and walks every kernel in text order: from an asm `global_load_dwordx4 v[a:b]` until the next asm `s_waitcnt vmcnt` (the waits of a
prefetch set come as one batch at the top of the next tile) or the loop's backward branch (the set is then waited for at the loop header),
no compiler-generated instruction may mention v[a:b].  Exit status 1 on a violation.

usage: python tools/check_opaque_loads.py [extra hipcc flags ...]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "ocrs_models_amd", "csrc", "det_mm.hip")


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "det_mm.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC", "-Wno-unused-result", *sys.argv[1:], "-S",
               "--cuda-device-only", SRC, "-o", out]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        txt = open(out).read().splitlines()
    labels = {}
    for i, line in enumerate(txt):
        m = re.match(r"^(\.LBB\w+):", line)
        if m:
            labels[m.group(1)] = i
    bad, kernels, name, in_asm, flight = [], set(), None, False, set()
    for i, line in enumerate(txt):
        s = line.strip()
        if re.match(r"^_Z\w+:", line):
            name, flight, in_asm = line.split(":")[0], set(), False
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        t = s.replace(",", " ").split()
        if not t or s.startswith((";", ".")):
            continue
        if in_asm:
            if t[0] == "global_load_dwordx4":
                flight |= regs(t[1])
                kernels.add(name)
            elif t[0] == "s_waitcnt":
                flight = set()
            continue
        if t[0] in ("s_branch", "s_cbranch_scc0", "s_cbranch_scc1", "s_cbranch_vccz", "s_cbranch_vccnz", "s_cbranch_execz", "s_cbranch_execnz"):
            if labels.get(t[1], 1 << 60) < i:  # backward branch: the set in flight is waited for at the loop header
                flight = set()
            continue
        if t[0] == "s_endpgm":
            flight = set()
            continue
        if flight:
            used = set().union(*[regs(x) for x in t[1:]]) if len(t) > 1 else set()
            if used & flight:
                bad.append((name, i, s, sorted(used & flight)))
    nk = len(kernels)
    for b in bad:
        print("VIOLATION", b)
    print(f"check_opaque_loads: {nk} kernels with opaque prefetch loads, {len(bad)} violations")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
