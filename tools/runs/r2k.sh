cd $GRAFT_REPO_ROOT
for a in "" "--no-roofline" "" "--no-roofline"; do python bench.py --no-crnn --no-cpu-baseline --no-fp32 $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a', d['value'], d['ms_per_step'])"; done
