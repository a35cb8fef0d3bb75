for tpb in 1 2 4; do
  export OCRS_FWD_TPB=$tpb
  bash tools/run_trace_step.sh >/dev/null 2>&1
  echo "FWD_TPB=$tpb"; grep "k_dwpw_fwd<bf16, 4" gpurun_out/trace_step.txt | head -20 | awk '{print $5,$NF}' | tr '\n' ';'; echo; grep "step span" gpurun_out/trace_step.txt;  grep "^k_bn_bwd_reduce\|^k_wgrad_partials" gpurun_out/trace_step.txt
done
