#!/bin/bash
# one gpurun call: HEADLINE-ONLY kernel-trace stats + PMC passes (FETCH_SIZE, WRITE_SIZE) of bench.py, the per-shape GEMM table; summaries into gpurun_out/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write
HL="--no-cpu-baseline --no-rooflines --no-ragged-workload"
rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof_stats -- python $R/bench.py --steps 5 --warmup 2 $HL > $O/prof_stats.log 2>&1 || { tail -20 $O/prof_stats.log; exit 1; }
echo stats done
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/prof_fetch -- python $R/bench.py --steps 1 --warmup 1 $HL > $O/prof_fetch.log 2>&1 || { tail -20 $O/prof_fetch.log; exit 1; }
echo fetch done
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/prof_write -- python $R/bench.py --steps 1 --warmup 1 $HL > $O/prof_write.log 2>&1 || { tail -20 $O/prof_write.log; exit 1; }
echo write done
cd $R
export SUMMARIZE_TOP=60
export SUMMARIZE_NOTE="rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rooflines --no-ragged-workload: the HEADLINE workload only (256 x 128 atoms x 256 tokens, default precision mode), 7 steps in the trace (2 warm-up + 5 timed)"
python profiles/summarize.py stats gpurun_out/prof_stats gpurun_out/kernel_stats.csv || { tail -5 $O/prof_stats.log; find $O/prof_stats | head; }
python profiles/summarize.py pmc gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/pmc_traffic.csv
python scratch/gemm_shapes.py > gpurun_out/gemm_shapes.txt 2>&1 || tail -5 gpurun_out/gemm_shapes.txt
find $O -type f -size +4M -delete
grep '"metric"' $O/prof_stats.log > $O/bench_profiled.json
cut -c1-200 $O/bench_profiled.json; tail -4 gpurun_out/gemm_shapes.txt
