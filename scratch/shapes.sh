for a in "--batch 64 --atoms 60 --tokens 100" "--batch 300 --atoms 128 --tokens 256" "--batch 64 --atoms 200 --tokens 256" "--batch 32 --atoms 250 --tokens 300 " "--batch 256 --atoms 128 --tokens 256 --ragged"; do
  timeout -k 10 300 python bench.py $a --steps 4 --warmup 2 --no-cpu-baseline 2>gpurun_out/shape.err | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', d['ms_per_step'], d['value'], d['losses_last_step'])" || { echo "FAILED $a"; tail -5 gpurun_out/shape.err; }
done
