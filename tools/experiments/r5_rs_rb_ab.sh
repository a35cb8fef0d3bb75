cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-crnn --no-fp32 --no-ref-style --no-ddp-probe --no-pmc --no-config1 --no-roofline --steps 30"
for rep in 1 2 3; do
for v in "OCRS_RS_RB=64" "OCRS_RS_RB=128" "OCRS_RS_RB=256" "OCRS_RS_RB=32"; do
  echo -n "$v  "; env $v $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done
