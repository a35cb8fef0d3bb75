cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gru_gpu.py tests/test_rec_gpu.py tests/test_train_loop_gpu.py tests/test_full_size_gpu.py -x -q -m gpu 2>&1 | tail -3
python tools/prof_crnn.py --steps 10 --warmup 3 2>&1 | tail -1 | cut -c1-160
