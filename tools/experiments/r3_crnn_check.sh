#!/bin/bash
# CRNN parity tests + CRNN bench line (+ the SQ counters of the block kernels for the profile set)
timeout 900 python -m pytest tests/test_rec_gpu.py tests/test_gru_gpu.py tests/test_train_loop_gpu.py -q -x -m gpu 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline --no-fp32 --no-ref-style --no-ddp-probe --no-config1 --no-gru-exact 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['crnn']; print('det', d['ms_per_step'], 'crnn', c['value'], c['ms_per_step'], c['roofline']['frac'])"
timeout 300 bash tools/run_trace_crnn.sh > /dev/null 2>&1; head -14 gpurun_out/crnn_stats.txt
