#!/bin/bash
# Cin = Cout = 32 block backward: two 256-thread workgroups per CU (default build) vs the one-workgroup form (c32old) vs without the merged dgrad pass (c32nomerge)
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/ocrs_models_amd/variants
timeout 900 python -m pytest tests/test_det_ops_gpu.py tests/test_det_bf16_layerwise_gpu.py -x -q 2>&1 | tail -3
for lib in "" c32old c32nomerge; do
  echo "== ${lib:-default}"
  if [ -n "$lib" ]; then export OCRS_LIB_PATH=$V/libocrs_hip_$lib.so; else unset OCRS_LIB_PATH; fi
  timeout 300 python tools/experiments/r5_c32_time.py 2>&1 | grep -v amdgpu.ids | tail -5
done
unset OCRS_LIB_PATH
bash tools/experiments/r5_det_ab.sh "n256" "old OCRS_LIB_PATH=$V/libocrs_hip_c32old.so" "nomerge OCRS_LIB_PATH=$V/libocrs_hip_c32nomerge.so"
