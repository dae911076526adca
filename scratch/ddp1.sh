set -e
timeout -k 10 300 python -m pytest tests/test_trainer_gpu.py -x -q -k "gradient_buckets" 2>&1 | tail -3
for v in 0 1 0 1; do
echo "NO_REDUCE_OVERLAP=$v"
MMDTI_NO_REDUCE_OVERLAP=$v MMDTI_FORCE_DDP=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', d['ms_per_step'])"
