#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_det_ops_gpu.py tests/test_det_model_gpu.py tests/test_edge_cases_gpu.py tests/test_det_bf16_layerwise_gpu.py -x -q 2>&1 | tail -3
bash tools/experiments/r5_det_ab.sh "default" "no16 OCRS_RSF_C16=0"
bash tools/run_trace_step.sh
