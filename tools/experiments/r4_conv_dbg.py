"""per-phase cycle counters of k_conv3x3_rows (variant built with -DR3_DBG=1): OCRS_LIB_PATH=ocrs_models_amd/variants/libocrs_hip_dbg.so"""
import ctypes, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_rec_gpu import _run, nhwc
dev = torch.device("cuda", 0)
for ci, co, H, W in [(128, 128, 8, 100), (128, 128, 16, 100)]:
    x = torch.randn(256, ci, H, W).to(dev); w = (torch.randn(co, ci, 3, 3) / math.sqrt(ci * 9)).to(dev); b = torch.randn(co).to(dev)
    r = _run(dev, torch.bfloat16, 256); r.P = {"w": w}; xs = nhwc(x, torch.bfloat16)
    for _ in range(3): r.conv(xs, w, b, True, True, H, W, 1, H, W)
    torch.cuda.synchronize()
    L = ctypes.CDLL(os.environ["OCRS_LIB_PATH"]); buf = (ctypes.c_longlong * 64)(); L.ocrs_conv_rows_dbg(buf)
    print(f"{ci}->{co} {H}x{W}: per wave [total, prologue, vmcnt wait, barrier wait, epilogue] (cycles of the 100 MHz? counter)")
    for wv in range(8): print("  wave", wv, [buf[wv * 8 + i] for i in range(5)])
