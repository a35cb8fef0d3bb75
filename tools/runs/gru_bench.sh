cd $GRAFT_REPO_ROOT
for m in 1 0; do
  OCRS_GRU_SEQ=$m python bench.py --steps 3 --warmup 2 --batch 4 --size 256 --no-cpu-baseline --no-fp32 > gpurun_out/gru_bench_$m.json 2> gpurun_out/gru_bench_$m.err
  python - gpurun_out/gru_bench_$m.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c=d["crnn"]; print("seq" , sys.argv[1], c["value"], c["ms_per_step"], json.dumps(c["roofline"].get("gru")), json.dumps({k:v for k,v in c.items() if "gru" in k or "other" in k})[:600])
PY
done
