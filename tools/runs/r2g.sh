cd $GRAFT_REPO_ROOT
bash tools/runs/variants.sh "$@" 2>&1 | grep -E "==|k_mm_bwd<(8, 8, false, false|8, 16|16, 8|16, 16, true)"
