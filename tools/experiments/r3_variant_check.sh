#!/bin/bash
# usage: r3_variant_check.sh NAME : block-kernel parity (op tests) and timings with ocrs_models_amd/variants/libocrs_hip_NAME.so against the regular build
V=ocrs_models_amd/variants/libocrs_hip_$1.so
echo "== regular"; timeout 300 python tools/experiments/r3_mm_time.py --fwd 2>&1 | grep -E "^bwd|^fwd"
echo "== $1"; OCRS_LIB_PATH=$V timeout 300 python tools/experiments/r3_mm_time.py --fwd 2>&1 | grep -E "^bwd|^fwd"
OCRS_LIB_PATH=$V timeout 900 python -m pytest tests/test_det_ops_gpu.py tests/test_det_bf16_layerwise_gpu.py -q -x -m gpu 2>&1 | tail -3
OCRS_LIB_PATH=$V timeout 400 python bench.py --no-cpu-baseline --no-crnn --no-fp32 --no-ref-style --no-ddp-probe --no-config1 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['passes']['block_bwd']['ms_per_step'], d['roofline']['passes']['block_fwd']['ms_per_step'])"
