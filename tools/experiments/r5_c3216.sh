#!/bin/bash
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/ocrs_models_amd/variants
timeout 900 python -m pytest tests/test_det_ops_gpu.py tests/test_det_model_gpu.py tests/test_edge_cases_gpu.py -x -q 2>&1 | tail -3
OCRS_LIB_PATH=$V/libocrs_hip_c3216.so timeout 900 python -m pytest tests/test_det_ops_gpu.py tests/test_det_bf16_layerwise_gpu.py -x -q 2>&1 | tail -3
bash tools/experiments/r5_det_ab.sh "default" "c3216 OCRS_LIB_PATH=$V/libocrs_hip_c3216.so"
export OCRS_LIB_PATH=$V/libocrs_hip_c3216.so
bash tools/run_trace_step.sh
