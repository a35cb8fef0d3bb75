# Collects every profile artefact of a round into gpurun_out/prof_<tag>/ (copy to profiles/ afterwards).  usage: bash tools/collect_profiles.sh r01_final
TAG=${1:-r01_final}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
# 1. un-profiled default bench line
timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
# 2. kernel stats of the same command (detection + CRNN), CPU baseline off
rm -rf gpurun_out/ks; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/ks.err
cp $(find gpurun_out/ks -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
# 3. per-launch trace of one detection step
bash tools/run_trace_step.sh > /dev/null 2>&1; cp gpurun_out/trace_step.txt $OUT/${TAG}_step_trace.txt
# 4. HBM traffic (two separate PMC passes)
bash tools/run_pmc_hbm.sh ${TAG}_pmc_hbm.csv > $OUT/pmc_hbm.log 2>&1; cp gpurun_out/${TAG}_pmc_hbm.csv $OUT/
# 5. SQ counters of the three big families
bash tools/run_pmc_sq.sh "k_dwpw_fwd<bf16, [12], 1, true>|k_pw_bwd2<(8|16), (8|16), (false|true)>|k_dw_bwd<bf16, [12], true>" > /dev/null 2>&1; cp gpurun_out/pmc_sq.txt $OUT/${TAG}_pmc_sq.txt
rm -rf gpurun_out/ks gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc_f gpurun_out/pmc_w
ls -la $OUT; tail -c 600 $OUT/${TAG}_bench.json
