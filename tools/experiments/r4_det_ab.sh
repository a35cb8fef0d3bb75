for i in 1 2; do
python bench.py --no-crnn --no-cpu-baseline --no-fp32 --no-ref-style --no-ddp-probe --no-config1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('regular', d['value'], d['ms_per_step'])"
OCRS_LIB_PATH=ocrs_models_amd/variants/libocrs_hip_mmnt.so python bench.py --no-crnn --no-cpu-baseline --no-fp32 --no-ref-style --no-ddp-probe --no-config1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nt     ', d['value'], d['ms_per_step'])"
done
