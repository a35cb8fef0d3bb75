cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6_full_tests.log 2>&1; tail -4 gpurun_out/r6_full_tests.log
timeout 900 python bench.py > gpurun_out/r6_bench.json 2> gpurun_out/r6_bench.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/r6_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('fp32_exact'), d['crnn']['value'], d['crnn']['ms_per_step'], d['crnn']['roofline']['frac'])
P
