cd $GRAFT_REPO_ROOT
for f in 1 0 1 0; do OCRS_FOLD_FIN=$f python bench.py --no-crnn --no-cpu-baseline --no-fp32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fold $f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['ms_per_step'])"; done
