for bpc in 0 4 2; do
  export OCRS_BPC=$bpc
  bash tools/run_trace_step.sh >/dev/null 2>&1
  echo "BPC=$bpc"; grep "k_dwpw_fwd<bf16" gpurun_out/trace_step.txt | head -25 | awk '{print $4,$5,$NF}' | tr '\n' ';'; echo; grep "step span\|^k_bce_fwd\|^k_topk" gpurun_out/trace_step.txt
done
