# families-per-step quick bench (detection only)
python bench.py --no-crnn --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_families_ms_warmup_step'])"
