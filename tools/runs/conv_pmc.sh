cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/cp1 gpurun_out/cp2
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/cp1 -- python tools/runs/gru_dbg.py > gpurun_out/cp1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC --output-format csv -d gpurun_out/cp2 -- python tools/runs/gru_dbg.py > gpurun_out/cp2.log 2>&1
python tools/pmc_summary.py gpurun_out/cp1 "$1"; python tools/pmc_summary.py gpurun_out/cp2 "$1"
rm -rf gpurun_out/cp1 gpurun_out/cp2
