# latency counters of the detection step: average VMEM / LDS instruction latency (SQ_INST_LEVEL_* / SQ_INSTS_*), L1->L2 and L2->HBM read latency
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pl1 gpurun_out/pl2 gpurun_out/pl3
ARGS="python bench.py --steps 1 --warmup 1 --no-crnn --no-cpu-baseline --no-roofline --no-fp32"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d gpurun_out/pl1 -- $ARGS > gpurun_out/pl1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pl2 -- $ARGS > gpurun_out/pl2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pl3 -- $ARGS > gpurun_out/pl3.log 2>&1
for d in pl1 pl2 pl3; do python tools/pmc_summary.py gpurun_out/$d "$1"; done > gpurun_out/pmc_lat.txt 2>&1
find gpurun_out/pl1 gpurun_out/pl2 gpurun_out/pl3 -name "*.csv" -size +2M -delete
tail -n 2 gpurun_out/pl1.log gpurun_out/pl2.log gpurun_out/pl3.log
