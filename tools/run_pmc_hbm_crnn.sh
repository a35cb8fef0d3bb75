# usage: bash tools/run_pmc_hbm_crnn.sh <out csv name under gpurun_out/>      (two separate PMC passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/cpmc_f gpurun_out/cpmc_w
ARGS="python tools/prof_crnn.py --steps 2 --warmup 1"
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/cpmc_f -- $ARGS > gpurun_out/cpmc_f.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/cpmc_w -- $ARGS > gpurun_out/cpmc_w.log 2>&1
python tools/pmc_hbm_crnn.py gpurun_out/cpmc_f gpurun_out/cpmc_w 4 gpurun_out/$1   # (bench_crnn: max(2, warmup) + steps = 4 steps)
find gpurun_out/cpmc_f gpurun_out/cpmc_w -name "*.csv" -size +1M -delete
