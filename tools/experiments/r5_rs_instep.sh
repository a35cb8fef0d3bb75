#!/bin/bash
# in-step durations of k_rs_bwd / k_mm_bwd<8, 8 (rocprofv3 kernel trace of 3 detection steps) for the libraries / settings given as "NAME ENV=..;ENV=.."
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  name=${spec%% *}; envs=${spec#* }; [ "$envs" = "$spec" ] && envs=""
  rm -rf gpurun_out/tr_$name
  env $(echo $envs | tr ';' ' ') rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_$name -- python bench.py --steps 3 --warmup 2 --no-crnn --no-cpu-baseline --no-roofline --no-fp32 --no-ref-style --no-ddp-probe --no-config1 --no-pmc > gpurun_out/tr_$name.log 2>&1
  python tools/trace_step.py gpurun_out/tr_$name > gpurun_out/tr_$name.txt 2>&1
  echo "[$name] $(grep -E '^k_rs_bwd|^k_mm_bwd<8, 8' gpurun_out/tr_$name.txt | tr -s ' ' | tr '\n' ';') $(grep 'step span' gpurun_out/tr_$name.txt)"
  find gpurun_out/tr_$name -name "*.csv" -delete
done
