cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_rec_gpu.py -x -q -k "conv_igemm or recognition" 2>&1 | tail -5
bash tools/runs/gru_bench.sh 2>&1 | cut -c1-100
