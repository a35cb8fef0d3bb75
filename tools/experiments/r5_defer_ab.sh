# A/B of the deferred second stage of the block backward on ONE box (round 5): default | in-kernel sums off for k_dw_bwd | in-kernel sums off | all off
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-crnn --no-fp32 --no-ref-style --no-ddp-probe --no-pmc --no-config1 --no-roofline --steps 20"
for rep in 1 2; do
for v in "" "OCRS_BWD_LAST_DW=0" "OCRS_BWD_LAST=0" "OCRS_BWD_DEFER=0"; do
  echo "== $v"; env $v $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done
