cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="python bench.py --dtype fp32 --steps 6 --warmup 2 --no-crnn --no-cpu-baseline --no-roofline --no-fp32 --no-ref-style --no-ddp-probe --no-config1 --no-pmc"
run() { echo "== $*"; env "$@" $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
run OCRS_RS32_RB=64
run OCRS_RS32_RB=32
run OCRS_RS32_RB=128
run OCRS_RS32_RB=256
run OCRS_RS32_RB=128 OCRS_RS32_CTW_RB=64 OCRS_RS32_CTF_RB=64
run OCRS_RS32_RB=128 OCRS_RS32_BPC=3
run OCRS_RS32_RB=64 OCRS_RS32_DUAL=0
run OCRS_RS32_RB=64 OCRS_OVERLAP=0
