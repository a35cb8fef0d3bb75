for cg in 4 2 1; do
  export OCRS_DW_CG=$cg
  bash tools/run_trace_step.sh >/dev/null 2>&1
  echo "DW_CG=$cg"; grep "k_dw_bwd<bf16" gpurun_out/trace_step.txt | head -25 | awk '{print $4,$5,$NF}' | tr '\n' ';'; echo; grep "step span" gpurun_out/trace_step.txt
done
