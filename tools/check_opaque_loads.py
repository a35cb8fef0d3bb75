"""Build-time guard for the hand-waited prefetch loads of csrc/det_mm.hip (gload16_opaque + wait_vm_tied).

hipcc believes an asm load has completed when the statement ends, so if its register allocator spills, copies or re-uses the destination
vector before the hand-written `s_waitcnt` the kernel reads or stores stale bytes.  This script disassembles det_mm.hip (hipcc -S, device
only) and walks every kernel in text order: from an asm `global_load_dwordx4 v[a:b]` until the next asm `s_waitcnt vmcnt` (the waits of a
prefetch set come as one batch at the top of the next tile) or the loop's backward branch (the set is then waited for at the loop header),
no compiler-generated instruction may mention v[a:b]; forward branches carry the in-flight set to their targets (text after an unconditional
branch is only reachable through its labels).  Exit status 1 on a violation.

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
    # OCRS_CHECK_SRC: another source of csrc/ with hand-waited asm loads (rec_conv3.hip: the A fragments of k_conv3x3_rows); default det_mm.hip
    global SRC
    if os.environ.get("OCRS_CHECK_SRC"):
        SRC = os.path.join(ROOT, "ocrs_models_amd", "csrc", os.environ["OCRS_CHECK_SRC"])
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "det_mm.s")
        # the build passes ITS compiler and flag list (OCRS_CHECK_HIPCC / OCRS_CHECK_FLAGS) so that the text checked here is the code it built
        hipcc = os.environ.get("OCRS_CHECK_HIPCC", "/opt/rocm/bin/hipcc")
        flags = os.environ.get("OCRS_CHECK_FLAGS", "--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -Wno-unused-result").split()
        cmd = [hipcc, *flags, *sys.argv[1:], "-S", "--cuda-device-only", SRC, "-o", out]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        txt = open(out).read().splitlines()
    bad, nk = check(txt)
    for b in bad:
        print("VIOLATION", b)
    print(f"check_opaque_loads ({os.path.basename(SRC)}): {nk} kernels with opaque prefetch loads, {len(bad)} violations")
    return 1 if bad else 0


def check(txt):
    """txt: lines of a `hipcc -S` dump.  Returns (violations, number of kernels with opaque loads); tests/test_isa_checker.py feeds it
    hand-written snippets."""
    labels = {}
    for i, line in enumerate(txt):
        m = re.match(r"^(\.LBB\w+):", line)
        if m:
            labels[m.group(1)] = i
    # natural loops = [label line, backward-branch line]; an edge that leaves a loop is an EXIT edge.  The tile loops issue no prefetch in
    # their last iteration (`if (t + step < end) issue(...)` and the loop condition are the same test), which a static walk cannot see: exit
    # edges therefore do not carry the in-flight set (the one assumption of this checker; the kernels also drain vmcnt right behind the loop).
    loops = []
    for i, line in enumerate(txt):
        t = line.strip().replace(",", " ").split()
        if t and (t[0] == "s_branch" or t[0].startswith("s_cbranch")) and len(t) > 1 and labels.get(t[1], 1 << 60) < i:
            loops.append((labels[t[1]], i))

    def leaves_loop(src, dst):  # (only loops that issue opaque loads: a backward branch to an out-of-line block is not a loop)
        return any(a <= src <= b and not (a <= dst <= b) for a, b in tile_loops)

    # text behind the last loop of a kernel = the code after the tile loop (flush / statistics): reached only with nothing in flight, for the
    # same reason (a block with a single tile leaves through the peeled first tile, whose prefetch is skipped by the same test)
    kernel_of, cur = {}, None
    for i, line in enumerate(txt):
        if re.match(r"^_Z\w+:", line):
            cur = line.split(":")[0]
        kernel_of[i] = cur
    asm_load_lines, in_a = [], False
    for i, line in enumerate(txt):
        st = line.strip()
        if st.startswith(";;#ASMSTART"):
            in_a = True
        elif st.startswith(";;#ASMEND"):
            in_a = False
        elif in_a and st.startswith("global_load_dwordx4"):
            asm_load_lines.append(i)
    tile_loops = [(a, b) for a, b in loops if any(a <= x <= b for x in asm_load_lines)]
    last_loop_end = {}  # kernel -> end of its last loop that issues opaque loads (= the tile loop)
    for a, b in loops:
        if any(a <= x <= b for x in asm_load_lines):
            last_loop_end[kernel_of[a]] = max(last_loop_end.get(kernel_of[a], 0), b)

    def next_is_branch(i):
        for line in txt[i + 1:i + 8]:
            u = line.strip()
            if u and not u.startswith((";", ".")):
                return u.split()[0] == "s_branch"
        return False

    CBR = ("s_cbranch_scc0", "s_cbranch_scc1", "s_cbranch_vccz", "s_cbranch_vccnz", "s_cbranch_execz", "s_cbranch_execnz")
    bad, kernels, name, in_asm = [], set(), None, False
    pending = {}                 # label -> [(registers in flight, known flags)] on forward edges into it
    # hipcc lowers a uniform if / else inside divergent code through FLAG registers: `s_mov_b64 s[a:b], -1 ... (then side) s_mov_b64 s[a:b], 0
    # ... s_andn2_b64 vcc, exec, s[a:b]; s_cbranch_vccnz SKIP_ELSE`.  Following both edges of that last branch would walk the else side with the
    # then side's loads in flight, so the walk keeps the flags whose value is a known constant on the current path (conservative: any other
    # write forgets the flag, joins keep only what all incoming edges agree on, loop headers forget everything) and resolves
    # `s_and[n2]_b64 vcc, exec, <flag>` + `s_cbranch_vcc[n]z` with them (exec != 0: with no lane enabled vector instructions write nothing).
    # The walk is path-sensitive in those flags: it carries a list of states [in flight, known flags, what vcc is known to be], merged when
    # their flags agree.
    loop_heads = {a for a, _ in loops}

    def sreg_range(tok):
        m = re.match(r"^s\[(\d+):(\d+)\]$", tok)
        if m:
            return set(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.match(r"^s(\d+)$", tok)
        return {int(m.group(1))} if m else set()

    def merged(sts, forget):
        out = {}
        for f, c in sts:
            c = {} if forget else c
            key = tuple(sorted(c.items()))
            if key in out:
                out[key][0] |= f
            else:
                out[key] = [set(f), dict(c), None]
        out = list(out.values())
        if len(out) > 16:  # (never seen; keep the walk bounded: one state, no flags)
            out = [[set().union(*[o[0] for o in out]), {}, None]]
        return out

    states = [[set(), {}, None]]  # empty list = not reachable by fall-through (behind an unconditional branch)
    for i, line in enumerate(txt):
        s = line.strip()
        if re.match(r"^_Z\w+:", line):
            name, in_asm, pending, states = line.split(":")[0], False, {}, [[set(), {}, None]]
            continue
        m = re.match(r"^(\.LBB\w+):", line)
        if m:
            states = merged([(f, c) for f, c, _ in states] + pending.pop(m.group(1), []), i in loop_heads)
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        t = s.replace(",", " ").split()
        if not t or s.startswith((";", ".")) or not states:
            continue
        if in_asm:
            if t[0] == "global_load_dwordx4":
                for st in states:
                    st[0] |= regs(t[1])
                kernels.add(name)
            elif t[0] == "s_waitcnt":
                for st in states:
                    st[0] = set()   # (the hand-written waits of a prefetch set come as one batch)
            continue
        if t[0] == "s_waitcnt" and "vmcnt(0)" in s:
            for st in states:
                st[0] = set()
            continue
        if t[0] == "s_cbranch_execz" and next_is_branch(i):
            # `s_cbranch_execz ELSE; s_branch JOIN` = hipcc's lowering of the end of the then-side of a UNIFORM if / else (structurizer "Flow"
            # block): the first edge is taken only with no lane enabled, in which case the vector instructions at ELSE write nothing -- not an
            # edge along which a destination register can be clobbered
            continue
        if t[0] == "s_branch" or t[0] in CBR:
            tgt = labels.get(t[1], 1 << 60)
            keep = []
            for st in states:
                taken, fall = True, t[0] != "s_branch"
                if t[0] in ("s_cbranch_vccnz", "s_cbranch_vccz") and st[2] is not None:
                    taken = (st[2] == "nonzero") == (t[0] == "s_cbranch_vccnz")
                    fall = not taken
                if taken and tgt > i and not leaves_loop(i, tgt):  # forward edge inside the same loop nest: the target inherits what is in flight here
                    pending.setdefault(t[1], []).append((set(st[0]), dict(st[1])))
                # (backward edge = the tile loop's back edge: the set in flight is waited for at the loop header before anything else)
                if fall:
                    if tgt < i:
                        st[0] = set()  # fall-through behind a loop's backward conditional branch = loop exit
                    keep.append(st)
            states = keep
            continue
        if t[0] == "s_endpgm":
            states = []
            continue
        for st in states:  # flag bookkeeping
            if len(t) > 1 and sreg_range(t[1]):
                st[1] = {k: v for k, v in st[1].items() if not (sreg_range(k) & sreg_range(t[1]))}
                if t[0] == "s_mov_b64" and len(t) == 3 and t[2] in ("0", "-1"):
                    st[1][t[1]] = int(t[2])
            if t[0] in ("s_and_b64", "s_andn2_b64") and len(t) == 4 and t[1] == "vcc" and t[2] == "exec" and t[3] in st[1]:
                st[2] = "nonzero" if (st[1][t[3]] == -1) == (t[0] == "s_and_b64") else "zero"
            elif "vcc" in s or re.match(r"^v_(cmpx?_\w+_e32|(add|sub|subrev|addc|subb|subbrev)_co_\w+_e32|div_scale|div_fmas)", t[0]):
                st[2] = None
        if i > last_loop_end.get(name, 1 << 60):
            for st in states:
                st[0] = set()
        used = set().union(*[regs(x) for x in t[1:]]) if len(t) > 1 else set()
        hit = set().union(*[st[0] for st in states]) & used
        if hit:
            bad.append((name, i, s, sorted(hit)))
    # warm-up loads (`global_load_dword` asm statements, OCRS_MM_WARM): nothing ever waits for them individually, so their destination
    # registers count as in flight from the first one to the end of the tile loop -- no compiler-emitted instruction may name them there
    warm, first = {}, {}
    in_a = False
    for i, line in enumerate(txt):
        st = line.strip()
        if st.startswith(";;#ASMSTART"):
            in_a = True
        elif st.startswith(";;#ASMEND"):
            in_a = False
        elif in_a and st.startswith("global_load_dword "):
            k = kernel_of[i]
            for r in regs(st.replace(",", " ").split()[1]):
                warm.setdefault(k, set()).add(r)
                first.setdefault((k, r), i)
    in_a = False
    for i, line in enumerate(txt):
        st = line.strip()
        if st.startswith(";;#ASMSTART"):
            in_a = True
        elif st.startswith(";;#ASMEND"):
            in_a = False
        k = kernel_of[i]
        if in_a or k not in warm or i > last_loop_end.get(k, 0):
            continue
        t = st.replace(",", " ").split()
        if not t or st.startswith((";", ".")) or re.match(r"^\.?\w+:", st):
            continue
        used = set().union(*[regs(x) for x in t[1:]]) if len(t) > 1 else set()
        hit = {r for r in used & warm[k] if first[(k, r)] < i}
        if hit:
            bad.append((k, i, st, sorted(hit), "warm-up destination"))
    unseen = {kernel_of[x] for x in asm_load_lines} - kernels
    for k in sorted(unseen):
        bad.append((k, 0, "the walk never reached this kernel's opaque loads (checker blind spot)", []))
    return bad, len(kernels)


if __name__ == "__main__":
    sys.exit(main())
