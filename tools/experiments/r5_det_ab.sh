#!/bin/bash
# detection train step, A / B on ONE box: bench.py (detection only) with the settings given as "NAME ENV=..;ENV=.." arguments, interleaved twice
cd $(dirname $0)/../..
ARGS="--steps 10 --warmup 3 --no-crnn --no-cpu-baseline --no-fp32 --no-ref-style --no-ddp-probe --no-config1"
for rep in 1 2; do
  for spec in "$@"; do
    name=${spec%% *}; envs=${spec#* }; [ "$envs" = "$spec" ] && envs=""
    line=$(env $(echo $envs | tr ';' ' ') python bench.py $ARGS 2>/dev/null | tail -1)
    python - "$name" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[2]); r = d.get("roofline", {})
print(f"[{sys.argv[1]:14s}] {d['ms_per_step']:.3f} ms  {d['value']:.0f} img/s  roofline.frac {r.get('frac')}  block_bwd_ms {r.get('ms_per_step_in_pass', r.get('pass_ms'))}")
PY
  done
done
