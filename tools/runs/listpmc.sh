cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|TA_[A-Z0-9_]*\|GRBM_[A-Z0-9_]*" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/pmc_list.txt
wc -l $GRAFT_REPO_ROOT/gpurun_out/pmc_list.txt
grep -E "LEVEL|WAIT|LATENCY|BUSY" $GRAFT_REPO_ROOT/gpurun_out/pmc_list.txt | tr '\n' ' '
