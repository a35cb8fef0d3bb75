#!/bin/bash
# one GPU call: parity of the detection kernels after an edit of csrc/det_mm.hip, then the block-kernel timings and a short bench line
mkdir -p gpurun_out/r3
timeout 1500 python -m pytest tests/test_det_ops_gpu.py tests/test_det_model_gpu.py tests/test_det_bf16_layerwise_gpu.py tests/test_edge_cases_gpu.py -q -x -m gpu 2>&1 | tail -5
timeout 300 python tools/experiments/r3_mm_time.py --fwd 2>&1 | tail -30
timeout 400 python bench.py --no-cpu-baseline --no-crnn --no-fp32 --no-ref-style --no-ddp-probe --no-config1 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline'])"
