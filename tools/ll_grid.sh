for B in 1024 65536 262144 524288 1048576; do
  timeout -k 10 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 300 --warmup 20 --bytes $B 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ranks 2 bytes $B us/op', round(d['ms_per_step']*1000,2), d['config']['plan'])" || exit 1
done
