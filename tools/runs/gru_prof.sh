cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gp -- python tools/runs/gru_dbg.py > gpurun_out/gp.log 2>&1
f=$(find gpurun_out/gp -name "*kernel_stats.csv" | head -1)
head -12 $f | cut -c1-160
rm -rf gpurun_out/gp
