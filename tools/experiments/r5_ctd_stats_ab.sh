cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_det_model_gpu.py tests/test_det_ops_gpu.py tests/test_edge_cases_gpu.py tests/test_full_size_gpu.py tests/test_det_bf16_layerwise_gpu.py -m gpu -x -q 2>&1 | grep -v "^$" | tail -3
B="python bench.py --no-cpu-baseline --no-crnn --no-fp32 --no-ref-style --no-ddp-probe --no-pmc --no-config1 --no-roofline --steps 30"
for rep in 1 2 3; do
for v in "OCRS_CTD_STATS=1" "OCRS_CTD_STATS=0"; do
  echo -n "$v  "; env $v $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done
bash tools/run_trace_step.sh > /dev/null 2>&1; grep "k_ctd\|k_bn_bwd_reduce" gpurun_out/trace_step.txt | head -8; grep -c grid= gpurun_out/trace_step.txt
