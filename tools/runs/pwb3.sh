cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_det_ops_gpu.py -x -q -m gpu -k "maxpool or dwpw_block" 2>&1 | tail -2
bash tools/run_trace_step.sh > /dev/null 2>&1; grep -n "k_pwb<.*true\|step span" gpurun_out/trace_step.txt | head -8
python bench.py --no-cpu-baseline --no-crnn --no-fp32 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['passes']['block_bwd'])"
