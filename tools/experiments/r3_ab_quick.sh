#!/bin/bash
# usage: r3_ab_quick.sh NAME [NAME2 ...] : backward / forward launch timings, regular build first and last (drift check), variants in between
echo "== regular"; timeout 300 python tools/experiments/r3_mm_time.py --fwd 2>&1 | grep -E "^bwd|^fwd"
for n in "$@"; do echo "== $n"; OCRS_LIB_PATH=ocrs_models_amd/variants/libocrs_hip_$n.so timeout 300 python tools/experiments/r3_mm_time.py --fwd 2>&1 | grep -E "^bwd|^fwd"; done
echo "== regular again"; timeout 300 python tools/experiments/r3_mm_time.py --fwd 2>&1 | grep -E "^bwd|^fwd"
