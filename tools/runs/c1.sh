cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_det_ops_gpu.py -x -q -m gpu -k "first_block" 2>&1 | tail -4
for v in "OCRS_C1V2=0" "OCRS_C1V2=1" "OCRS_C1V2_FWD_BPC=4 OCRS_C1V2_BWD_BPC=3" "OCRS_C1V2_FWD_BPC=6 OCRS_C1V2_BWD_BPC=5" "OCRS_C1V2_FWD_BPC=16 OCRS_C1V2_BWD_BPC=8" "OCRS_C1V2_FWD_BPC=32 OCRS_C1V2_BWD_BPC=16"; do
env $v python tools/runs/c1_bench.py 2>&1 | tail -1
done
timeout 900 python -m pytest tests/test_full_size_gpu.py tests/test_det_model_gpu.py -x -q -m gpu 2>&1 | tail -3
