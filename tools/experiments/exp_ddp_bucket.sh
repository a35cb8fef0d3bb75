# DDP overhead at 1 rank (RCCL all-reduce forced) for different bucket sizes; compare with the plain run
cd $GRAFT_REPO_ROOT
echo "plain: $(python bench.py --steps 10 --warmup 3 --no-crnn --no-cpu-baseline --no-roofline 2>/dev/null | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')"
for bb in 262144 1048576 4194304 67108864; do
  echo "bucket=$bb: $(OCRS_DDP_FORCE=1 OCRS_DDP_BUCKET_BYTES=$bb python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 10 --warmup 3 --no-crnn --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')"
done
