# usage: env.sh VAR  -> alternates VAR=1 / VAR=0 three times
for k in 1 2 3; do
 for v in 1 0; do
  export $1=$v
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1=$v', d['ms_per_step'])"
 done
done
