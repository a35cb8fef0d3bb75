#!/bin/bash
# OCRS_MM_REV experiment: step time with / without alternating tile directions (same box, interleaved), and parity of the reversed order
B="python bench.py --no-cpu-baseline --no-crnn --no-fp32 --no-ref-style --no-ddp-probe --no-config1 --no-roofline"
for i in 1 2; do
  for r in 0 1; do echo -n "REV=$r: "; OCRS_MM_REV=$r timeout 300 $B 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"; done
done
OCRS_MM_REV=1 timeout 900 python -m pytest tests/test_det_ops_gpu.py tests/test_det_bf16_layerwise_gpu.py -q -x -m gpu 2>&1 | tail -2
