#!/bin/bash
for k in 1 2 3; do
 for v in old new; do
  MMDTI_HIP_LIB=$PWD/scratch/ab/lib_$v.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'])"
 done
done
