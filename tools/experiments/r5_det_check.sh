#!/bin/bash
# detection GPU tests + a per-launch trace of one step
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu -k "det or edge or bce or loss or golden or graph or train_loop or full_size" 2>&1 | tail -4
bash tools/run_trace_step.sh
