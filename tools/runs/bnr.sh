cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in 8 4 2 3; do
export OCRS_BNR_BPC=$v
bash tools/run_trace_step.sh > /dev/null 2>&1; echo "bpc $v"; grep "k_bn_bwd_reduce" gpurun_out/trace_step.txt | head -4 | awk '{print $NF}' | tr '\n' ' '; grep "step span" gpurun_out/trace_step.txt
done
