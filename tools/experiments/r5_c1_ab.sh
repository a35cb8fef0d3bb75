cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-crnn --no-fp32 --no-ref-style --no-ddp-probe --no-pmc --no-config1 --no-roofline --steps 30"
for rep in 1 2 3; do
for v in "OCRS_C1_FUSE=1" "OCRS_C1_FUSE=0"; do
  echo -n "$v  "; env $v $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done
bash tools/run_trace_step.sh > /dev/null 2>&1; grep "k_rs_bwd\|k_c1_\|k_bn_bwd_finalize" gpurun_out/trace_step.txt | head; tail -1 gpurun_out/trace_step.txt
