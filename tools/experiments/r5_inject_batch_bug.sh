#!/bin/bash
# Test of the tests: a build whose first-block forward kernel writes image n's output to image n ^ 1 must FAIL the
# distinct-data full-batch test and PASS the replicated-tile test (which cannot see it: all images are equal there).
# Round 6: the bug is no longer an #ifdef in the product source -- this script patches a COPY of det_c1.hip (the two stores of k_c1_fwd2) and builds
# a variant library from it (tools/build_variant.sh, VARIANT_SRC_DIR).
cd $(dirname $0)/../..
D=$(mktemp -d); cp ocrs_models_amd/csrc/*.h $D/
sed -e 's|uplane\[((long)it.n \* H + it.h0 + q)|uplane[((long)((it.n ^ 1) < N ? (it.n ^ 1) : it.n) * H + it.h0 + q)|' \
    -e 's|store8(z + (((long)it.n \* H + it.h0 + q)|store8(z + (((long)((it.n ^ 1) < N ? (it.n ^ 1) : it.n) * H + it.h0 + q)|' ocrs_models_amd/csrc/det_c1.hip > $D/det_c1.hip
if cmp -s $D/det_c1.hip ocrs_models_amd/csrc/det_c1.hip; then echo "patch did not apply"; exit 1; fi
VARIANT_SRC_DIR=$D tools/build_variant.sh batchbug "" det_c1.hip
rm -rf $D
export OCRS_LIB_PATH=$PWD/ocrs_models_amd/variants/libocrs_hip_batchbug.so
python -m pytest tests/test_full_size_gpu.py -q -x -k "distinct_tiles" 2>&1 | tail -3
python -m pytest tests/test_full_size_gpu.py -q -x -k "replicated_tile" 2>&1 | tail -3
