cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc1 gpurun_out/pmc2
ARGS="python bench.py --steps 1 --warmup 1 --no-crnn --no-cpu-baseline --no-roofline --no-fp32 --no-ref-style --no-ddp-probe --no-config1"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc1 -- $ARGS > gpurun_out/pmc1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc2 -- $ARGS > gpurun_out/pmc2.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc1 "$1" > gpurun_out/pmc_sq.txt 2>&1
python tools/pmc_summary.py gpurun_out/pmc2 "$1" >> gpurun_out/pmc_sq.txt 2>&1
find gpurun_out/pmc1 gpurun_out/pmc2 -name "*.csv" -size +2M -delete
tail -n 2 gpurun_out/pmc1.log gpurun_out/pmc2.log
