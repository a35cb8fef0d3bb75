cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu -k "bce or loss or golden or detection" 2>&1 | tail -4
bash tools/run_trace_step.sh
