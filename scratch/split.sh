for k in 1 2; do
for v in 1 2 4; do
  MMDTI_SPLIT_TOWER1=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split $v', d['ms_per_step'])"
done; done
