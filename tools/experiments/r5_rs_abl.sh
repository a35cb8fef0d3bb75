#!/bin/bash
cd $(dirname $0)
V=$PWD/../../ocrs_models_amd/variants
run() { TAG="$1" "${@:2}" python r5_rs_time.py 2>&1 | grep "^\["; }
python r5_rs_check.py | tail -1
for b in 768 1024; do run rb64_blocks$b env OCRS_RS_BLOCKS=$b; done
run rb32 env OCRS_RS_RB=32
run rb128 env OCRS_RS_RB=128
for v in $VARIANTS; do run $v env OCRS_LIB_PATH=$V/libocrs_hip_$v.so; done
