#!/bin/bash
# 16-channel instantiations of the row-streaming block backward (variant build -DOCRS_RS_16): op tests on the variant, then the step A / B and a trace
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/ocrs_models_amd/variants
OCRS_LIB_PATH=$V/libocrs_hip_rs16.so timeout 900 python -m pytest tests/test_det_ops_gpu.py tests/test_det_bf16_layerwise_gpu.py tests/test_det_model_gpu.py -x -q 2>&1 | tail -4
bash tools/experiments/r5_det_ab.sh "default" "rs16 OCRS_LIB_PATH=$V/libocrs_hip_rs16.so"
export OCRS_LIB_PATH=$V/libocrs_hip_rs16.so
bash tools/run_trace_step.sh
