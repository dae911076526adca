#!/bin/bash
# one gpurun call: rocprofv3 kernel-trace stats of `bench.py <args>`; summary -> gpurun_out/<tag>_kernel_stats.csv
#   scratch/prof_stats.sh <tag> <bench.py arguments...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
rm -rf $O/prof_$tag
rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof_$tag -- python $R/bench.py "$@" > $O/prof_$tag.log 2>&1 || { tail -20 $O/prof_$tag.log; exit 1; }
cd $R
python profiles/summarize.py stats gpurun_out/prof_$tag gpurun_out/${tag}_kernel_stats.csv
grep '"metric"' $O/prof_$tag.log > $O/${tag}_bench.json
find $O/prof_$tag -type f -size +4M -delete
head -45 gpurun_out/${tag}_kernel_stats.csv | cut -c1-200
