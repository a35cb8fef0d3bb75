#!/bin/bash
# Test of the tests: a build whose first-block forward kernel writes image n's output to image n ^ 1 must FAIL the
# distinct-data full-batch test and PASS the replicated-tile test (which cannot see it: all images are equal there).
cd $(dirname $0)/../..
tools/build_variant.sh batchbug "-DOCRS_INJECT_BATCH_BUG=1" det_c1.hip
export OCRS_LIB_PATH=$PWD/ocrs_models_amd/variants/libocrs_hip_batchbug.so
python -m pytest tests/test_full_size_gpu.py -q -x -k "distinct_tiles" 2>&1 | tail -3
python -m pytest tests/test_full_size_gpu.py -q -x -k "replicated_tile" 2>&1 | tail -3
