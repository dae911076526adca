#!/bin/bash
# one gpurun call: kernel-trace stats + PMC passes (FETCH_SIZE, WRITE_SIZE, read-request counters) of bench.py; summaries into gpurun_out/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write $O/prof_rdreq $O/prof_rdreq2
rocprofv3 -L > $O/counters_all.txt 2>&1
grep -oE "TCC_EA0_RDREQ[A-Za-z0-9_]*|TCC_EA0_RD[A-Za-z0-9_]*|TCP_TCC_READ_REQ[A-Za-z0-9_]*|TCC_REQ[A-Za-z0-9_]*|TCC_READ[A-Za-z0-9_]*" $O/counters_all.txt | sort -u > $O/counters_rd.txt
rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof_stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_stats.log 2>&1 || { tail -20 $O/prof_stats.log; exit 1; }
echo stats done
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/prof_fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-rooflines --no-ragged-workload > $O/prof_fetch.log 2>&1 || { tail -20 $O/prof_fetch.log; exit 1; }
echo fetch done
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/prof_write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-rooflines --no-ragged-workload > $O/prof_write.log 2>&1 || { tail -20 $O/prof_write.log; exit 1; }
echo write done
rocprofv3 --output-format csv --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $O/prof_rdreq -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-rooflines --no-ragged-workload > $O/prof_rdreq.log 2>&1 || { tail -5 $O/prof_rdreq.log; echo "rdreq pass failed (continuing)"; }
rocprofv3 --output-format csv --kernel-trace --pmc TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $O/prof_rdreq2 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-rooflines --no-ragged-workload > $O/prof_rdreq2.log 2>&1 || { tail -5 $O/prof_rdreq2.log; echo "rdreq 64/128 pass failed (continuing)"; rm -rf $O/prof_rdreq2; }
cd $R
python profiles/summarize.py stats gpurun_out/prof_stats gpurun_out/kernel_stats.csv || { tail -5 $O/prof_stats.log; find $O/prof_stats | head; }
python profiles/summarize.py pmc gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/pmc_traffic.csv
if [ -d gpurun_out/prof_rdreq2 ]; then python profiles/summarize.py rdreq gpurun_out/prof_rdreq gpurun_out/pmc_rdreq.csv gpurun_out/prof_rdreq2 || echo "no rdreq summary";
else python profiles/summarize.py rdreq gpurun_out/prof_rdreq gpurun_out/pmc_rdreq.csv || echo "no rdreq summary"; fi
find $O -type f -size +4M -delete
grep '"metric"' $O/prof_stats.log > $O/bench_profiled.json
cut -c1-400 $O/bench_profiled.json
