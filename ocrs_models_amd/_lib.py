"""ctypes binding of libocrs_hip.so (the C ABI declared in include/ocrs_hip.h).

The product path has NO CPU fallback: if the library is missing this raises, and every
entry point raises RuntimeError on a non-zero status.
"""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OCRS_LIB_PATH") or os.path.join(_HERE, "libocrs_hip.so")  # (OCRS_LIB_PATH: measurement builds, tools/build_variant.sh)

_T = {"p": ctypes.c_void_p, "i": ctypes.c_int, "l": ctypes.c_long, "f": ctypes.c_float, "d": ctypes.c_double, "s": ctypes.c_void_p}

HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "ocrs_hip.h")


def parse_header(path: str = HEADER_PATH):
    """include/ocrs_hip.h is the single source of truth for the ABI: name -> (restype, arg signature)."""
    import re

    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    sigs = {}
    for m in re.finditer(r"^(int|long)\s+(ocrs_\w+)\(([^;]*)\);", text, re.M | re.S):
        res, name, args = m.group(1), m.group(2), m.group(3).strip()
        sig = ""
        if args not in ("", "void"):
            for a in args.split(","):
                a = a.strip()
                if "hipStream_t" in a:
                    sig += "s"
                elif "*" in a:
                    sig += "p"
                elif a.startswith("long"):
                    sig += "l"
                elif a.startswith("int"):
                    sig += "i"
                elif a.startswith("float"):
                    sig += "f"
                elif a.startswith("double"):
                    sig += "d"
                else:
                    raise RuntimeError(f"unparsed argument {a!r} of {name}")
        sigs[name] = ("i" if res == "int" else "l", sig)
        ARG_NAMES[name] = [re.sub(r"\[.*", "", a.strip().split()[-1].lstrip("*")) for a in args.split(",")] if args not in ("", "void") else []
    return sigs


ARG_NAMES = {}  # ocrs_* -> parameter names in ABI order (for tools that read call arguments by name: bench.py's byte model)
SIGNATURES = parse_header()

_ERR = {1: "bad argument", 2: "HIP launch/runtime error"}


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m ocrs_models_amd.build` "
                "(there is no CPU/eager fallback in the product path)"
            )
        self._dll = ctypes.CDLL(LIB_PATH)
        for name, (res, sig) in SIGNATURES.items():
            fn = getattr(self._dll, name)
            fn.restype = _T[res]
            fn.argtypes = [_T[c] for c in sig]
            setattr(self, "_raw_" + name, fn)
            setattr(self, name[5:], self._wrap(name, fn, res, sig))

    # name (without the ocrs_ prefix) -> list of (start_event, end_event, args); None = timing off.
    # Used by bench.py to time the dominant kernel family live, with events on the launch stream.
    timing = None
    # name -> list of (first, end, args): launch-index range in the C library's dispatch-timestamp recorder (ocrs_prof_enable / _read: no stream
    # events, the kernels stay back to back); None = off.  Used for the launches inside bench.py's TIMED region.
    prof = None

    def _wrap(self, name, fn, res, sig):
        has_stream = sig.endswith("s")
        short = name[5:]
        owner = self

        trace = bool(os.environ.get("OCRS_TRACE"))  # debugging aid: print every entry point before the launch and synchronise after it

        def call(*args):
            rec = None
            if trace:
                print("[ocrs]", name, [a if not isinstance(a, int) or a < (1 << 32) else hex(a) for a in args], flush=True)
            if owner.timing is not None and short in owner.timing:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rec = (e0, e1, args)
            pr0 = owner._dll.ocrs_prof_count() if (owner.prof is not None and short in owner.prof) else -1
            if has_stream:
                r = fn(*args, torch.cuda.current_stream().cuda_stream)
            else:
                r = fn(*args)
            if pr0 >= 0:
                owner.prof[short].append((pr0, owner._dll.ocrs_prof_count(), args))
            if rec is not None:
                rec[1].record()
                owner.timing[short].append(rec)
            if trace:
                torch.cuda.synchronize()
            if res == "i" and sig and r != 0:
                raise RuntimeError(f"{name} failed: {_ERR.get(r, r)}")
            return r

        call.__name__ = name
        return call


_lib = None


def lib() -> _Lib:
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib


def ptr(t):
    """device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
