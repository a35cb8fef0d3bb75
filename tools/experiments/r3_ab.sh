#!/bin/bash
# usage: r3_ab.sh NAME : block-kernel timings of the regular build (A) and of variants/libocrs_hip_NAME.so (B), then parity + bench of the REGULAR build
V=ocrs_models_amd/variants/libocrs_hip_$1.so
echo "== A regular"; timeout 300 python tools/experiments/r3_mm_time.py --fwd 2>&1 | grep -E "^bwd|^fwd"
echo "== B $1"; OCRS_LIB_PATH=$V timeout 300 python tools/experiments/r3_mm_time.py --fwd 2>&1 | grep -E "^bwd|^fwd"
echo "== A again"; timeout 300 python tools/experiments/r3_mm_time.py 2>&1 | grep -E "^bwd total"
timeout 1500 python -m pytest tests/test_det_ops_gpu.py tests/test_det_model_gpu.py tests/test_det_bf16_layerwise_gpu.py tests/test_edge_cases_gpu.py -q -x -m gpu 2>&1 | tail -3
timeout 400 python bench.py --no-cpu-baseline --no-crnn --no-fp32 --no-ref-style --no-ddp-probe --no-config1 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['passes']['block_bwd']['ms_per_step'], d['roofline']['passes']['block_fwd']['ms_per_step'])"
