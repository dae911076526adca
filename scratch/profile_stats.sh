#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; rm -rf $O/prof_stats
rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof_stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_stats.log 2>&1 || { tail -20 $O/prof_stats.log; exit 1; }
cd $R && python profiles/summarize.py stats gpurun_out/prof_stats gpurun_out/kernel_stats.csv
find $O -type f -size +4M -delete
grep '"metric"' $O/prof_stats.log | cut -c1-200
