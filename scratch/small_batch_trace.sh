#!/bin/bash
# kernel trace of the B=32 ragged step, eager and graph-replayed: how much of the step is kernel time, how many kernels, what is the gap
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
for mode in eager graph; do
  rm -rf $O/prof_sb_$mode
  extra=""; [ $mode = graph ] && extra="--graph"
  rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof_sb_$mode -- python $R/bench.py --batch 32 --ragged $extra --steps 20 --warmup 5 --no-cpu-baseline --no-rooflines > $O/prof_sb_$mode.log 2>&1 || { tail -5 $O/prof_sb_$mode.log; exit 1; }
  grep -o "ms_per_step\": [0-9.]*" $O/prof_sb_$mode.log | sed "s/^/$mode: /"
done
cd $R
python - <<'PY'
import csv, glob, os
for mode in ("eager", "graph"):
    f = sorted(glob.glob(f"gpurun_out/prof_sb_{mode}/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last 20 steps: take the final 40 % of kernels as steady state
    n = len(rows); tail = rows[int(n * 0.6):]
    dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tail)
    span = int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])
    # per-stream serial gaps are not visible here; report totals
    print(f"{mode}: kernels in window {len(tail)}  sum of kernel durations {dur/1e6:.2f} ms  wall span {span/1e6:.2f} ms  busy fraction {dur/span:.2f}  mean kernel {dur/len(tail)/1e3:.2f} us  mean span per kernel {span/len(tail)/1e3:.2f} us")
PY
find $O -type f -size +4M -delete
