#!/bin/bash
# usage: r5_ab_libs.sh VARIANT...   -- det op tests on the default build, then the detection step A / B: default build vs ocrs_models_amd/variants/libocrs_hip_VARIANT.so
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/ocrs_models_amd/variants
timeout 900 python -m pytest tests/test_det_ops_gpu.py tests/test_det_bf16_layerwise_gpu.py tests/test_det_model_gpu.py -x -q 2>&1 | tail -3
specs=("default")
for v in "$@"; do specs+=("$v OCRS_LIB_PATH=$V/libocrs_hip_$v.so"); done
bash tools/experiments/r5_det_ab.sh "${specs[@]}"
