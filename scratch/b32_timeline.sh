#!/bin/bash
# per-queue timeline of the B=32 step: kernel time, gaps and the longest kernels on each HIP stream over the last steps of the trace
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; rm -rf $O/prof_b32tl
rocprofv3 --output-format csv --kernel-trace -d $O/prof_b32tl -- python $R/scratch/b32_loop.py 30 ${1:-0} > $O/prof_b32tl.log 2>&1 || { tail -5 $O/prof_b32tl.log; exit 1; }
tail -1 $O/prof_b32tl.log
cd $R
python - <<'PY'
import csv, glob, os, collections
f = sorted(glob.glob("gpurun_out/prof_b32tl/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps are delimited by adam_kernel
ad = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
lo, hi = ad[-11], ad[-1]          # ten steps
win = rows[lo + 1:hi + 1]
span = (int(win[-1]["End_Timestamp"]) - int(rows[lo]["End_Timestamp"])) / 10e3
print(f"ten steps: {len(win)/10:.0f} kernels/step, {span:.0f} us/step wall between adam kernels")
byq = collections.defaultdict(list)
for r in win: byq[r["Queue_Id"]].append(r)
out = []
for q, rs in byq.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 10e3
    gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rs, rs[1:])]
    small = sum(g for g in gaps if 0 < g < 30000) / 10e3
    out.append(f"queue {q}: {len(rs)/10:.0f} kernels/step, busy {busy:.0f} us/step, gaps<30us total {small:.0f} us/step (mean {small*10/max(1,sum(1 for g in gaps if 0<g<30000)):.1f} us)")
print("\n".join(out))
# one step's sequence on each queue with start offsets
step = rows[ad[-2] + 1:ad[-1] + 1]
t0 = int(step[0]["Start_Timestamp"])
with open("gpurun_out/b32_timeline.txt", "w") as fo:
    fo.write("\n".join(out) + "\n")
    for r in step:
        fo.write(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f} q{r['Queue_Id']} {r['Kernel_Name'][:90]}\n")
PY
find $O/prof_b32tl -type f -size +4M -delete
