for bpc in 0 4 5 6; do
  export OCRS_BPC=$bpc
  bash tools/run_trace_step.sh >/dev/null 2>&1
  echo "BPC=$bpc"; grep "^k_dwpw_fwd\|^k_dw_bwd" gpurun_out/trace_step.txt | awk '{s+=$NF} END {print "fwd+dw_bwd total us:", s}'; grep "step span" gpurun_out/trace_step.txt
done
