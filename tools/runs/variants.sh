# usage: bash tools/runs/variants.sh name1 name2 ...   (per-kernel stats of each variant lib; "default" = the regular build)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = default ]; then
    bash tools/runs/prof_stats.sh var_$v > /dev/null 2>&1
  else
    bash tools/runs/prof_stats.sh var_$v OCRS_LIB_PATH=$GRAFT_REPO_ROOT/ocrs_models_amd/variants/libocrs_hip_$v.so > /dev/null 2>&1
  fi
  echo "== $v: $(head -c 250 gpurun_out/var_${v}_bench.json | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' ')"
  grep -E "k_mm_fwd|k_mm_bwd|finalize_parts" gpurun_out/var_${v}_kstats.txt | head -40
done
