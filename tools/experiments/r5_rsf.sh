#!/bin/bash
# row-streaming forward (k_rs_fwd): tests, then the step with it off / on at 2 / 3 / 4 workgroups per CU, then a trace
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/ocrs_models_amd/variants
timeout 1200 python -m pytest tests/test_det_ops_gpu.py tests/test_det_bf16_layerwise_gpu.py tests/test_det_model_gpu.py tests/test_edge_cases_gpu.py tests/test_full_size_gpu.py -x -q -k "not crnn and not recognition" 2>&1 | tail -4
bash tools/experiments/r5_det_ab.sh "rsf3" "off OCRS_RSF=0" "rsf2 OCRS_LIB_PATH=$V/libocrs_hip_rsf2.so" "rsf4 OCRS_LIB_PATH=$V/libocrs_hip_rsf4.so"
bash tools/run_trace_step.sh
