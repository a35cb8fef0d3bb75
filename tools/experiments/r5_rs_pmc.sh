#!/bin/bash
# HBM-side counters of k_rs_bwd / k_mm_bwd at level-0 size (separate PMC passes, kernel trace only)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/rs_pmc; rm -rf $O; mkdir -p $O
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_EA0_[A-Z0-9_]*\|TCC_HIT[A-Z_a-z]*\|TCC_MISS[A-Z_a-z]*\|TCC_REQ[A-Z_a-z]*" | sort -u | tr '\n' ' ' > $O/avail.txt
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  n=$(echo $pass | tr ' ' '+')
  for rs in 1 0; do
    OCRS_RS=$rs TAG=rs$rs timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/p_${n}_$rs -- python tools/experiments/r5_rs_time.py > $O/log_${n}_$rs.txt 2>&1
    python tools/pmc_summary.py $O/p_${n}_$rs "k_rs_bwd|k_mm_bwd<" >> $O/summary.txt 2>&1
  done
done
find $O -name "*.csv" -size +1M -delete
cat $O/avail.txt; echo; cat $O/summary.txt
