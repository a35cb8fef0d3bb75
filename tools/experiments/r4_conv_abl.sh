#!/bin/bash
# ablation builds of rec_conv3.hip (tools/build_variant.sh ablN "-DR3_ABL=N" rec_conv3.hip): time per call of the two main shapes
for v in "" 1 2 4 7 16 23; do
  if [ -z "$v" ]; then L=""; else L=ocrs_models_amd/variants/libocrs_hip_abl$v.so; fi
  echo "== ablation mask ${v:-0}"
  OCRS_LIB_PATH=$L R4_SHAPES=2 timeout 300 python tools/experiments/r4_conv_time.py 2>&1 | grep "N=256" | cut -c1-75
done
