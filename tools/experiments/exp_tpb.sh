# deep-level k_pw_bwd: minimum tiles per block (OCRS_WGRAD_TPB) sweep, per-kernel totals from a kernel trace
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for tpb in 4 2 1; do
  export OCRS_WGRAD_TPB=$tpb
  bash tools/run_trace_step.sh >/dev/null 2>&1
  echo "== TPB=$tpb"; grep "^k_pw_bwd<\|^k_wgrad_partials_reduce\|step span" gpurun_out/trace_step.txt
done
