# Round 5: whole-step A/B of launch-geometry knobs on ONE box (detection only, 30 timed steps, two interleaved repetitions)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-crnn --no-fp32 --no-ref-style --no-ddp-probe --no-pmc --no-config1 --no-roofline --steps 30"
for rep in 1 2; do
for v in "OCRS_X=0" "OCRS_C1V2_BWD_BPC=4" "OCRS_C1V2_BWD_BPC=7" "OCRS_C1V2_FWD_BPC=5" "OCRS_C1V2_FWD_BPC=12" "OCRS_HEADL_BPC=4" "OCRS_HEADL_BPC=16" "OCRS_WGRAD_TPB=8" "OCRS_BNR_BPC=4" "OCRS_DW_BPC=6"; do
  echo -n "$v  "; env $v $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done
