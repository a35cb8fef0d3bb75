import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from r5_rs_check import run_case
out, _ = run_case(8, 8, 1, 33, 61, True)
print(os.environ.get("OCRS_RS"), "S1", out["gsum"][:8].numpy().round(2), "T", out["gsum"][8:].numpy().round(2))
