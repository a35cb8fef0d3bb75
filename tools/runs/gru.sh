cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gru_gpu.py -x -q 2>&1 | tail -15
