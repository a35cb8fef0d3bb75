cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_det_model_gpu.py tests/test_edge_cases_gpu.py tests/test_train_loop_gpu.py -x -q 2>&1 | tail -3
for o in 1 0; do OCRS_OVERLAP=$o python bench.py --no-crnn --no-cpu-baseline --no-fp32 --no-roofline 2>/dev/null | head -c 200; echo; done
