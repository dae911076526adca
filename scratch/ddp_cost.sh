#!/bin/bash
# what the data-parallel machinery costs on ONE rank (forced 1-rank process group): kernel stats with / without it
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
for v in 1 0; do
  rm -rf $O/prof_ddp$v
  MMDTI_FORCE_DDP=$v rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof_ddp$v -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rooflines > $O/prof_ddp$v.log 2>&1 || { tail -5 $O/prof_ddp$v.log; exit 1; }
  grep -o '"ms_per_step": [0-9.]*' $O/prof_ddp$v.log | head -1
done
cd $R
python - <<'PY'
import csv, glob, os
def load(d):
    f = sorted(glob.glob(f"gpurun_out/{d}/**/*kernel_stats.csv", recursive=True), key=os.path.getsize)[-1]
    return {r["Name"][:90]: (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(f))}
a, b = load("prof_ddp1"), load("prof_ddp0")
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0, 0.0)); cb, tb = b.get(k, (0, 0.0))
    rows.append((ta - tb, k, ca, cb, ta, tb))
rows.sort(reverse=True)
print("largest differences, ms over 7 steps (ddp - plain):")
for d, k, ca, cb, ta, tb in rows[:14]: print(f"  {d:8.2f}  calls {ca:5d}/{cb:5d}  {ta:8.2f}/{tb:8.2f}  {k}")
print("  total", sum(r[0] for r in rows))
PY
