"""Build libocrs_hip.so (hand-written gfx950 kernels + C ABI) in-tree with hipcc.

    python -m ocrs_models_amd.build          # incremental
hipcc cross-compiles for gfx950 without a GPU present.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libocrs_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC", "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    jobs = []
    for src in sources():
        s, o = os.path.join(CSRC, src), os.path.join(OBJ, src + ".o")
        if force or _newer(s, o) or any(_newer(h, o) for h in headers):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [hipcc, *FLAGS, "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stdout}\n{r.stderr}")

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(cc, jobs))
    for asm_src in ("det_mm.hip", "rec_conv3.hip", "det_rs.hip"):
        if not any(os.path.basename(s) == asm_src for s, _ in jobs):
            continue
        # the hand-waited asm loads of det_mm.hip (tile prefetch) and rec_conv3.hip (A fragments) are only valid if hipcc left their
        # destination registers alone until the wait
        chk = os.path.join(os.path.dirname(HERE), "tools", "check_rs_loads.py" if asm_src == "det_rs.hip" else "check_opaque_loads.py")
        if not os.path.exists(chk):
            knob = {"det_mm.hip": "OCRS_MM_FULL=0", "rec_conv3.hip": "OCRS_CONV_ROWS=0", "det_rs.hip": "OCRS_RS=0 OCRS_RSF=0"}[asm_src]
            print(f"WARNING: {os.path.basename(chk)} not found -- {asm_src}'s hand-waited asm loads were NOT verified against this "
                  f"compiler's register allocation (build from the repository tree, or run with {knob} to use the compiler-waited kernels)",
                  file=sys.stderr, flush=True)
        else:
            r = subprocess.run([sys.executable, chk], capture_output=True, text=True,
                               env={**os.environ, "OCRS_CHECK_HIPCC": hipcc, "OCRS_CHECK_FLAGS": " ".join(FLAGS), "OCRS_CHECK_SRC": asm_src})
            if verbose:
                print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr, flush=True)
            if r.returncode != 0:
                os.remove(os.path.join(OBJ, asm_src + ".o"))
                raise RuntimeError(f"{os.path.basename(chk)}: hipcc touched an in-flight asm-load register in {asm_src}:\n" + r.stdout[-3000:])
    objs = [os.path.join(OBJ, src + ".o") for src in sources()]
    if force or jobs or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


TORCH_OPS_SRC = os.path.join(CSRC, "torch_ops.cpp")
TORCH_OPS_LIB = os.path.join(HERE, "libocrs_torch_ops.so")


def build_torch_ops(force: bool = False, verbose: bool = True) -> str:
    """libocrs_torch_ops.so: TORCH_LIBRARY(ocrs, ...) registration of C-ABI entry points as torch.ops.ocrs.* (host-only C++, links
    libocrs_hip.so through an $ORIGIN rpath and the libtorch of the running interpreter)."""
    import torch

    header = os.path.join(os.path.dirname(HERE), "include", "ocrs_hip.h")
    if not (force or _newer(TORCH_OPS_SRC, TORCH_OPS_LIB) or _newer(header, TORCH_OPS_LIB) or _newer(LIB, TORCH_OPS_LIB)):
        return TORCH_OPS_LIB
    tdir = os.path.dirname(torch.__file__)
    cmd = [_hipcc(), "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", f"-I{tdir}/include", f"-I{tdir}/include/torch/csrc/api/include",
           "-I/opt/rocm/include", "-x", "c++", TORCH_OPS_SRC, "-o", TORCH_OPS_LIB, f"-L{tdir}/lib", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10",
           "-lc10_hip", f"-L{HERE}", "-locrs_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"torch ops build failed:\n{r.stdout}\n{r.stderr}")
    return TORCH_OPS_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    print(build_torch_ops(force="--force" in sys.argv))
