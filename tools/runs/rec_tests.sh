cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_rec_gpu.py tests/test_full_size_gpu.py tests/test_edge_cases_gpu.py tests/test_train_loop_gpu.py -x -q -k "rec or crnn or CRNN or Rec" 2>&1 | tail -6
bash tools/runs/gru_bench.sh 2>&1 | cut -c1-330
