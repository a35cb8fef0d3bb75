# usage: bash tools/run_pmc_hbm.sh <out csv name under gpurun_out/>      (two separate PMC passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
ARGS="python bench.py --steps 2 --warmup 1 --no-crnn --no-cpu-baseline --no-roofline --no-fp32 --no-ref-style --no-ddp-probe --no-config1"
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- $ARGS > gpurun_out/pmc_f.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- $ARGS > gpurun_out/pmc_w.log 2>&1
python tools/pmc_hbm.py gpurun_out/pmc_f gpurun_out/pmc_w 3 gpurun_out/$1
find gpurun_out/pmc_f gpurun_out/pmc_w -name "*.csv" -size +1M -delete
