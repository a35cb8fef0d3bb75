# Collects every profile artefact of a round into gpurun_out/prof_<tag>/ (copy to profiles/ afterwards).  usage: bash tools/collect_profiles.sh r03_final
TAG=${1:-r03_final}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
# 1. un-profiled default bench line (what the driver runs)
timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
# 2. kernel stats of the same command (detection + CRNN), CPU baseline off
rm -rf gpurun_out/ks; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fp32 --no-gru-exact --no-rec-config5 --no-ref-style --no-ddp-probe --no-config1 > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/ks.err
cp $(find gpurun_out/ks -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
# 3. per-launch trace of one detection step
bash tools/run_trace_step.sh > /dev/null 2>&1; cp gpurun_out/trace_step.txt $OUT/${TAG}_step_trace.txt
# 3b. per-launch trace of one detection step in the fp32 parity mode (round 6: the row-streaming fp32 kernels of csrc/det_rs32.hip)
rm -rf gpurun_out/trace32; timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace32 -- python bench.py --dtype fp32 --steps 3 --warmup 2 --no-crnn --no-cpu-baseline --no-roofline --no-fp32 --no-ref-style --no-ddp-probe --no-config1 --no-pmc > gpurun_out/trace32.log 2>&1
python tools/trace_step.py gpurun_out/trace32 > $OUT/${TAG}_fp32_step_trace.txt 2>&1; rm -rf gpurun_out/trace32
# 4. HBM traffic (two separate PMC passes, kernel-trace only)
bash tools/run_pmc_hbm.sh ${TAG}_pmc_hbm.csv > $OUT/pmc_hbm.log 2>&1; cp gpurun_out/${TAG}_pmc_hbm.csv $OUT/
# 5. SQ counters of the matrix-core / row-streaming block kernels (every instantiation of the step)
bash tools/run_pmc_sq.sh "k_mm_bwd<|k_rs_bwd<|k_mm_fwd<|k_rs_fwd<" > /dev/null 2>&1; cp gpurun_out/pmc_sq.txt $OUT/${TAG}_pmc_sq.txt
# 6. CRNN: per-kernel time, and the MFMA-busy counter of the convolution kernels (separate PMC pass)
bash tools/run_trace_crnn.sh > /dev/null 2>&1; cp gpurun_out/crnn_stats.txt $OUT/${TAG}_crnn_kernel_stats.txt
rm -rf gpurun_out/pmc_mfma; timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma -- python tools/prof_crnn.py --steps 2 --warmup 1 > gpurun_out/pmc_mfma.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_mfma "k_conv_igemm<bf16|k_conv3x3_c128|k_conv3x3_rows|k_conv3x3_tile|k_conv3x3_wgrad_tr|k_gemm_x3|k_wgrad_gemm_x3|k_gru_s" > $OUT/${TAG}_crnn_pmc_mfma.txt 2>&1
# 7. per-launch trace of one CRNN step
rm -rf gpurun_out/trace_crnn; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_crnn -- python tools/prof_crnn.py --steps 3 --warmup 2 > gpurun_out/trace_crnn.log 2>&1
python tools/trace_step.py gpurun_out/trace_crnn k_conv0_fwd > $OUT/${TAG}_crnn_step_trace.txt 2>&1
# 8. CRNN: HBM bytes and achieved GB/s per kernel (two separate PMC passes)
bash tools/run_pmc_hbm_crnn.sh ${TAG}_crnn_pmc_hbm.csv > $OUT/${TAG}_crnn_pmc_hbm.txt 2>&1; cp gpurun_out/${TAG}_crnn_pmc_hbm.csv $OUT/
rm -rf gpurun_out/cpmc_f gpurun_out/cpmc_w
rm -rf gpurun_out/ks gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_mfma gpurun_out/trace_crnn
ls -la $OUT; head -c 700 $OUT/${TAG}_bench.json; echo; tail -3 $OUT/${TAG}_step_trace.txt; tail -2 $OUT/${TAG}_pmc_hbm.csv; head -30 $OUT/${TAG}_crnn_pmc_mfma.txt
