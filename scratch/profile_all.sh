#!/bin/bash
# one gpurun call: stats trace + two PMC passes of bench.py, summaries into gpurun_out/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write
rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof_stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_stats.log 2>&1 || { tail -20 $O/prof_stats.log; exit 1; }
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/prof_fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_fetch.log 2>&1 || { tail -20 $O/prof_fetch.log; exit 1; }
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/prof_write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_write.log 2>&1 || { tail -20 $O/prof_write.log; exit 1; }
cd $R
du -sh $O/* | sort -h | tail -5
python profiles/summarize.py stats gpurun_out/prof_stats gpurun_out/kernel_stats.csv || { tail -5 $O/prof_stats.log; find $O/prof_stats | head; }
python profiles/summarize.py pmc gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/pmc_traffic.csv
find $O -type f -size +4M -delete
grep '"metric"' $O/prof_stats.log | cut -c1-300
