cd $GRAFT_REPO_ROOT
for v in "" noload nols nocomp; do
  echo "=== variant: ${v:-base}"
  if [ -n "$v" ]; then export OCRS_LIB_PATH=$GRAFT_REPO_ROOT/ocrs_models_amd/variants/libocrs_hip_$v.so; else unset OCRS_LIB_PATH; fi
  timeout 300 python tools/r3/mm_time.py 2>&1 | grep -v amdgpu.ids
done
