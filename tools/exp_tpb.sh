for tpb in 2 8 32; do
  export OCRS_WGRAD_TPB=$tpb
  bash tools/run_trace_step.sh >/dev/null 2>&1
  echo "TPB=$tpb"; grep "k_pw_bwd<bf16, \(64\|128\|256\)" gpurun_out/trace_step.txt | head -12 | awk '{print $3,$4,$5,$6,$NF}' | tr '\n' ';'; echo; grep "step span" gpurun_out/trace_step.txt
done
