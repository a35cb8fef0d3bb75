"""Build-time guard for the hand-waited prefetch loads of csrc/det_rs.hip (bload16_opaque + wait_vm).

hipcc believes an asm load has completed when the statement ends; if its register allocator spills, copies or re-uses a destination vector
before the hand-written wait, the kernel reads or stores stale bytes.  The waits of det_rs.hip name the registers they release in a
comment (`s_waitcnt vmcnt(N) ; releases v[a:b]`), so the check is an exact forward data-flow over the kernel's control-flow graph:
    in flight  :=  destination registers of asm `buffer_load_dword[x4]`, until an asm wait that names them (or an asm `s_waitcnt vmcnt(0)`)
    violation  :=  any compiler-generated instruction that mentions a register in flight, or any scratch access in a kernel with such loads
(union at joins, iterated to a fixed point).  Kernels whose asm loads name ACCUMULATION registers as destinations (k_rs_fwd: a[0:3], a[4:7]) are
checked for the stronger property that no compiler-generated instruction mentions those registers at all.  Exit status 1 on a violation.

usage: python tools/check_rs_loads.py [extra hipcc flags ...]      (OCRS_CHECK_HIPCC / OCRS_CHECK_FLAGS / OCRS_CHECK_SRC as check_opaque_loads.py)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def vregs(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out |= set(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        out.add(int(a))
    return out


def aregs(text):
    out = set()
    for a, b in re.findall(r"\ba\[(\d+):(\d+)\]", text):
        out |= set(range(int(a), int(b) + 1))
    for a in re.findall(r"\ba(\d+)\b", text):
        out.add(int(a))
    return out


def check(txt):
    """txt: lines of a `hipcc -S` dump -> (violations, kernels with opaque loads)."""
    bad, nk = [], 0
    starts = [i for i, l in enumerate(txt) if re.match(r"^_Z\w+:", l)]
    for si, s in enumerate(starts):
        e = next((i for i in range(s, len(txt)) if txt[i].startswith(".Lfunc_end")), len(txt) - 1)  # (blocks may follow the first s_endpgm)
        name = txt[s].split(":")[0]
        # instruction list with asm flags
        ins, in_asm, labels = [], False, {}
        for i in range(s + 1, e + 1):
            st = txt[i].strip()
            m = re.match(r"^(\.LBB\w+):", st)
            if m:
                labels[m.group(1)] = len(ins)
                continue
            if st.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if st.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not st or st.startswith((";", ".")):
                continue
            ins.append((i, st, in_asm))
        if not any(a and t.startswith("buffer_load_dword") for _, t, a in ins):
            continue
        nk += 1
        # prefetch sets in accumulation registers named in the asm text (k_rs_fwd): hipcc must not use those registers for anything, anywhere
        mine = set()
        for _, t, a in ins:
            if a and t.startswith("buffer_load_"):
                mine |= aregs(t.split(",")[0])
        if mine:
            for ln, t, a in ins:
                hit = aregs(t.split(";")[0]) & mine
                if hit and not a:
                    bad.append(f"{name}: line {ln + 1}: a{sorted(hit)} (an asm prefetch destination) used by compiler-generated code: {t}")
        n = len(ins)
        succ = [[] for _ in range(n)]
        for k, (_, t, _) in enumerate(ins):
            op = t.replace(",", " ").split()
            if op[0] == "s_endpgm":
                continue
            if op[0] == "s_branch":
                succ[k].append(labels[op[1]])
                continue
            if op[0].startswith("s_cbranch"):
                succ[k].append(labels[op[1]])
            if k + 1 < n:
                succ[k].append(k + 1)
        inflight = [None] * n  # set at entry of instruction k
        inflight[0] = frozenset()
        work = [0]
        while work:
            k = work.pop()
            cur = set(inflight[k])
            _, t, is_asm = ins[k]
            code = t.split(";")[0]
            if is_asm and code.startswith("buffer_load_dword"):
                cur |= vregs(code.split(",")[0])
            elif is_asm and code.startswith("s_waitcnt"):
                if re.search(r"vmcnt\(0\)", code):
                    cur = set()
                elif "releases" in t:
                    cur -= vregs(t.split("releases")[1])
            out = frozenset(cur)
            for j in succ[k]:
                new = out if inflight[j] is None else inflight[j] | out
                if new != inflight[j]:
                    inflight[j] = new
                    work.append(j)
        for k, (ln, t, is_asm) in enumerate(ins):
            if inflight[k] is None:
                continue
            code = t.split(";")[0]
            if "scratch_" in code:
                bad.append(f"{name}: line {ln + 1}: scratch access in a kernel with hand-waited loads: {t}")
            if is_asm:
                continue
            hit = vregs(code) & inflight[k]
            if hit:
                bad.append(f"{name}: line {ln + 1}: v{sorted(hit)} touched while an asm load into it is in flight: {t}")
    return bad, nk


def main():
    src = os.path.join(ROOT, "ocrs_models_amd", "csrc", os.environ.get("OCRS_CHECK_SRC", "det_rs.hip"))
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        hipcc = os.environ.get("OCRS_CHECK_HIPCC", "/opt/rocm/bin/hipcc")
        flags = os.environ.get("OCRS_CHECK_FLAGS", "--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -Wno-unused-result").split()
        subprocess.run([hipcc, *flags, *sys.argv[1:], "-S", "--cuda-device-only", src, "-o", out], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        txt = open(out).read().splitlines()
    bad, nk = check(txt)
    for b in bad[:40]:
        print("VIOLATION", b)
    print(f"check_rs_loads ({os.path.basename(src)}): {nk} kernels with hand-waited loads, {len(bad)} violations")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
