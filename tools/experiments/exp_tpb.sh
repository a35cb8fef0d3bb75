for cfg in "8 0" "8 1" "2 1" "4 1"; do
  set -- $cfg
  export OCRS_WGRAD_TPB=$1 OCRS_NOFLUSH=$2
  bash tools/run_trace_step.sh >/dev/null 2>&1
  echo "TPB=$1 NOFLUSH=$2"; grep "k_pw_bwd<bf16, \(64\|128\|256\)\|k_pw_bwd<bf16, 32, 64" gpurun_out/trace_step.txt | head -13 | awk '{print $4,$5,$NF}' | tr '\n' ';'; echo; grep "step span" gpurun_out/trace_step.txt
done
