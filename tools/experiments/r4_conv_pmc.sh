#!/bin/bash
# SQ counters of k_conv3x3_rows (and the old k_conv3x3_c128 for reference) on the two main shapes; $1 = optional variant library
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export OCRS_LIB_PATH=$1 R4_SHAPES=1
rm -rf gpurun_out/cpmc1 gpurun_out/cpmc2
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/cpmc1 -- python tools/experiments/r4_conv_time.py > gpurun_out/cpmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC --output-format csv -d gpurun_out/cpmc2 -- python tools/experiments/r4_conv_time.py > gpurun_out/cpmc2.log 2>&1
python tools/pmc_summary.py gpurun_out/cpmc1 "k_conv3x3" 
python tools/pmc_summary.py gpurun_out/cpmc2 "k_conv3x3"
tail -n 3 gpurun_out/cpmc1.log gpurun_out/cpmc2.log
find gpurun_out/cpmc1 gpurun_out/cpmc2 -name "*.csv" -size +1M -delete
