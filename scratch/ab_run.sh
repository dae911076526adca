#!/bin/bash
# same-box A/B of two builds of the library: scratch/libmmdti_old.so (built by hand from an older source) vs the in-tree one
# usage: bash scratch/ab_run.sh <tag> <bench args...>
tag=$1; shift
for i in 1 2; do
  MMDTI_HIP_LIB=$PWD/scratch/libmmdti_old.so python bench.py --no-cpu-baseline --no-rooflines "$@" > gpurun_out/${tag}_old_$i.json 2>/dev/null || exit 1
  python bench.py --no-cpu-baseline --no-rooflines "$@" > gpurun_out/${tag}_new_$i.json 2>/dev/null || exit 1
done
grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/${tag}_*.json
