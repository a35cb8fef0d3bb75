cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc3
ARGS="python bench.py --steps 1 --warmup 1 --no-crnn --no-cpu-baseline --no-roofline --no-fp32 --no-ref-style --no-ddp-probe --no-config1"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc3 -- $ARGS > gpurun_out/pmc3.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc3 "$1" > gpurun_out/pmc_lds.txt 2>&1
find gpurun_out/pmc3 -name "*.csv" -size +2M -delete
tail -n 3 gpurun_out/pmc3.log
