#!/bin/bash
# concurrency profile of one headline step: how much of the step has 0 / 1 / >= 2 kernels in flight, and what runs in the stretches with one
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; rm -rf $O/prof_tl
rocprofv3 --output-format csv --kernel-trace -d $O/prof_tl -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-rooflines --no-ragged-workload > $O/prof_tl.log 2>&1 || { tail -5 $O/prof_tl.log; exit 1; }
cd $R
python - <<'PY'
import csv, glob, os, collections
f = sorted(glob.glob("gpurun_out/prof_tl/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
lo, hi = ad[-2], ad[-1]
step = rows[lo + 1:hi + 1]
t0 = int(rows[lo]["End_Timestamp"]); t1 = int(step[-1]["End_Timestamp"])
ev = []
for r in step:
    ev.append((int(r["Start_Timestamp"]), 1, r)); ev.append((int(r["End_Timestamp"]), -1, r))
ev.sort(key=lambda e: (e[0], e[1]))
cover = collections.Counter(); active = {}; last = t0
single = collections.Counter()
for t, d, r in ev:
    n = len(active)
    cover[min(n, 3)] += t - last
    if n == 1:
        single[next(iter(active.values()))["Kernel_Name"][:70]] += t - last
    last = t
    if d == 1: active[id(r)] = r
    else: active.pop(id(r), None)
tot = t1 - t0
out = [f"step {tot/1e6:.2f} ms: idle {cover[0]/1e6:.2f}  one kernel {cover[1]/1e6:.2f}  two {cover[2]/1e6:.2f}  three+ {cover[3]/1e6:.2f} ms"]
byq = collections.defaultdict(float)
for r in step: byq[r["Queue_Id"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
out.append("busy per queue (ms): " + ", ".join(f"q{q}: {v:.2f}" for q, v in sorted(byq.items())))
out.append("time with exactly ONE kernel in flight, by kernel (ms):")
for k, v in single.most_common(25): out.append(f"  {v/1e6:7.3f}  {k}")
open("gpurun_out/step_timeline.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
find $O/prof_tl -type f -size +4M -delete
