"""standalone times of the (Cin, Cout) block backward at level-0 size for the library in OCRS_LIB_PATH (variants) and settings of OCRS_RS / OCRS_RS_BLOCKS"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from r5_rs_check import run_case
tag = os.environ.get("TAG", "")
cases = [tuple(int(v) for v in c.split(",")) for c in os.environ.get("RS_CASES", "8,8").split(";")]
for ci, co in cases:
    for g2 in (False, True):
        _, t = run_case(ci, co, 32, 1024, 1024, g2, timing=True)
        px = 32 * 1024 * 1024
        print(f"[{tag}] bwd ({ci},{co}) g2={int(g2)}: {t:8.1f} us  {px * 4 * (ci + co) / t / 1e6:5.2f} TB/s algorithmic", flush=True)
