#!/bin/bash
# GPU idle time inside the timed steps of bench.py: rocprofv3 kernel trace -> union of kernel intervals over all streams -> gaps
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
rm -rf $O/prof_gaps
rocprofv3 --output-format csv --kernel-trace -d $O/prof_gaps -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rooflines > $O/prof_gaps.log 2>&1 || { tail -5 $O/prof_gaps.log; exit 1; }
cd $R
python - <<'PY'
import csv, glob, os
f = sorted(glob.glob("gpurun_out/prof_gaps/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)[-1]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# the timed region = the last 5 steps: find adam kernels (one per step)
adam = [e for e in ev if "adam_kernel" in e[2]]
t0, t1 = adam[-6][1], adam[-1][1]
ev = [e for e in ev if e[0] >= t0 and e[1] <= t1]
busy, cur_s, cur_e, gaps = 0, None, None, []
for s, e, n in ev:
    if cur_e is None: cur_s, cur_e, last = s, e, n
    elif s <= cur_e:
        if e > cur_e: cur_e, last = e, n
    else:
        busy += cur_e - cur_s; gaps.append((s - cur_e, last, n)); cur_s, cur_e, last = s, e, n
busy += cur_e - cur_s
wall = t1 - t0
print(f"5 steps: wall {wall/1e6:.2f} ms, GPU busy {busy/1e6:.2f} ms = {100*busy/wall:.1f} %, idle {(wall-busy)/1e6/5:.3f} ms per step in {len(gaps)//5} gaps per step")
gaps.sort(reverse=True)
for g, a, b in gaps[:12]:
    print(f"  {g/1e3:7.1f} us  after {a[:60]}  before {b[:60]}")
import collections
c = collections.Counter()
for g, a, b in gaps: c[(a[:50], b[:50])] += g
print("by kernel pair (us per step):")
for (a, b), g in c.most_common(10): print(f"  {g/5e3:7.1f}  {a} -> {b}")
PY
