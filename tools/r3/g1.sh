#!/bin/bash
# first GPU call of round 3: bf16 parity experiment + baseline bench
mkdir -p gpurun_out/r3
python tools/r3/bf16_parity.py > gpurun_out/r3/bf16_parity.txt 2>&1
tail -70 gpurun_out/r3/bf16_parity.txt
python bench.py > gpurun_out/r3/bench0.json 2> gpurun_out/r3/bench0.err
tail -c 3000 gpurun_out/r3/bench0.json
