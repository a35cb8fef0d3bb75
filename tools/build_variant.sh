#!/bin/bash
# usage: tools/build_variant.sh NAME "<extra hipcc flags>" [source.hip ...]   -> ocrs_models_amd/variants/libocrs_hip_NAME.so
# Re-compiles the given sources (default: det_mm.hip) with the extra -D flags and links them with the other objects of the regular build;
# select at run time with OCRS_LIB_PATH=ocrs_models_amd/variants/libocrs_hip_NAME.so (measurement knob, see _lib.py).
# VARIANT_SRC_DIR=<dir>: take the listed sources (and the headers) from <dir> instead of csrc/ (patched copies, e.g. tools/experiments/r5_inject_batch_bug.sh).
set -e
NAME=$1; FLAGS=$2; shift 2
SRCS=${@:-det_mm.hip}
ROOT=$(cd $(dirname $0)/.. && pwd)
C=$ROOT/ocrs_models_amd/csrc; V=$ROOT/ocrs_models_amd/variants; mkdir -p $V/obj_$NAME
OBJS=""
for f in $C/*.hip; do
  b=$(basename $f)
  if echo " $SRCS " | grep -q " $b "; then
    if [ -n "$VARIANT_SRC_DIR" ] && [ -f "$VARIANT_SRC_DIR/$b" ]; then f=$VARIANT_SRC_DIR/$b; fi
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -Wno-unused-result $FLAGS -c $f -o $V/obj_$NAME/$b.o
    OBJS="$OBJS $V/obj_$NAME/$b.o"
  else
    OBJS="$OBJS $C/build/$b.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libocrs_hip_$NAME.so $OBJS
rm -rf $V/obj_$NAME
echo built $V/libocrs_hip_$NAME.so
