#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of one bench.py run into the small summaries committed under profiles/.

  python profiles/summarize.py stats  <rocprof-dir> profiles/r01_bench_kernel_stats.csv
  python profiles/summarize.py pmc    <fetch-dir> <write-dir> profiles/r01_bench_pmc_hbm_traffic.csv

`stats`: the *_kernel_stats.csv of `rocprofv3 --kernel-trace --stats`, template arguments kept, top 40 rows.
`pmc`:   per-kernel means of FETCH_SIZE and WRITE_SIZE (two separate passes, TCC has no room for both) converted to HBM
         bytes per launch as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE (KB) is doubled (128-B requests are
         tallied at 64 B), WRITE_SIZE (KB) is taken as is.
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def find(d, pat):
    hits = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True), key=os.path.getsize)
    if not hits:
        sys.exit(f"no {pat} under {d}")
    return hits[-1]


def demangle(name):
    """rocprofv3 leaves kernels whose template arguments include _Float16 / __bf16 mangled (its demangler predates DF16_ / DF16b).
    Only what this library's kernel names need: _ZN5mmdti<len><name>I<args>E... with int (Li9E), bool (Lb1E), float (f),
    _Float16 (DF16_) and __bf16 (DF16b) arguments."""
    m = re.match(r"_ZN5mmdti(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    base, rest = name[m.end():m.end() + n], name[m.end() + n:]
    if not rest.startswith("I"):
        return "mmdti::" + base
    rest, args = rest[1:], []
    while rest and not rest.startswith("E"):
        for pat, fn in ((r"Li(\d+)E", lambda g: g.group(1)), (r"Lb([01])E", lambda g: "true" if g.group(1) == "1" else "false"),
                        (r"DF16_", lambda g: "_Float16"), (r"DF16b", lambda g: "__bf16"), (r"f", lambda g: "float")):
            g = re.match(pat, rest)
            if g:
                args.append(fn(g)); rest = rest[g.end():]
                break
        else:
            return name
    return f"mmdti::{base}<{', '.join(args)}>"


def short(name):
    name = re.sub(r"\(.*$", "", name)            # drop the argument list, keep template args
    return demangle(name.replace("void ", "").strip())


def stats(src, dst):
    rows = list(csv.DictReader(open(find(src, "*kernel_stats.csv"))))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open(dst, "w") as f:
        f.write("# " + os.environ.get("SUMMARIZE_NOTE", "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline (every step of the run is in the "
                                                  "trace: warm-up, timed, the mixed-length workload, the per-family roofline steps)") + "\n")
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "percent", "min_us", "max_us"])
        for r in rows[:int(os.environ.get("SUMMARIZE_TOP", "40"))]:
            w.writerow([short(r["Name"]), r["Calls"], f'{float(r["TotalDurationNs"]) / 1e6:.3f}', f'{float(r["AverageNs"]) / 1e3:.2f}',
                        f'{float(r["Percentage"]):.2f}', f'{float(r["MinNs"]) / 1e3:.2f}', f'{float(r["MaxNs"]) / 1e3:.2f}'])


def counter_means(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    per_dispatch = defaultdict(float)
    names = {}
    for r in csv.DictReader(open(find(d, "*counter_collection.csv"))):
        if r["Counter_Name"] != counter:
            continue
        key = (r.get("Dispatch_Id") or r.get("Correlation_Id"))
        per_dispatch[key] += float(r["Counter_Value"])
        names[key] = short(r["Kernel_Name"])
    for key, v in per_dispatch.items():
        a = acc[names[key]]
        a[0] += v
        a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def pmc(fetch_dir, write_dir, dst):
    fe, wr = counter_means(fetch_dir, "FETCH_SIZE"), counter_means(write_dir, "WRITE_SIZE")
    rows = []
    for k in fe:
        f_kb, n = fe[k]
        w_kb = wr.get(k, (0.0, 0))[0]
        rows.append((k, n, f_kb, w_kb, 2 * f_kb * 1024 + w_kb * 1024))
    rows.sort(key=lambda r: -r[4] * r[1])
    with open(dst, "w") as f:
        f.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- python bench.py --steps 1 --warmup 1 "
                "--no-cpu-baseline --no-rooflines --no-ragged-workload (the headline step only); hbm_bytes_per_launch = 2*FETCH_SIZE_KB*1024 + WRITE_SIZE_KB*1024 (gfx950 correction of MI355X_MICROARCH.md); "
                "steps_in_trace=2 (the warm-up step + the step); every kernel of the trace is listed\n")
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "FETCH_SIZE_KB_mean", "WRITE_SIZE_KB_mean", "hbm_bytes_per_launch"])
        for r in rows:
            w.writerow([r[0], r[1], f"{r[2]:.1f}", f"{r[3]:.1f}", f"{r[4]:.0f}"])


def rdreq(src, dst, src2=None):
    """Second view of the read traffic (is FETCH_SIZE x 2 right for the tiled pair reads?): the L2's memory-side
    read REQUEST counters.  TCC_EA0_RDREQ counts every request, TCC_EA0_RDREQ_32B the 32-byte ones; FETCH_SIZE is
    RDREQ x 64 B (MI355X_MICROARCH.md, HBM): if the remaining requests are 128 B wide the bytes read are
    32 * n32 + 128 * (n - n32), if 64 B wide 32 * n32 + 64 * (n - n32) -- both are listed.  With a second pass
    (``src2``: TCC_EA0_RDREQ_64B_sum, TCC_EA0_RDREQ_128B_sum) the split is measured and ``read_bytes_by_size`` =
    32 n32 + 64 n64 + 128 n128 is the read traffic itself."""
    n_all, n32 = counter_means(src, "TCC_EA0_RDREQ_sum"), counter_means(src, "TCC_EA0_RDREQ_32B_sum")
    n64 = n128 = {}
    if src2:
        n64, n128 = counter_means(src2, "TCC_EA0_RDREQ_64B_sum"), counter_means(src2, "TCC_EA0_RDREQ_128B_sum")
    rows = []
    for k, (v, n) in n_all.items():
        v32 = n32.get(k, (0.0, 0))[0]
        v64, v128 = n64.get(k, (None, 0))[0], n128.get(k, (None, 0))[0]
        by_size = None if v64 is None or v128 is None else 32 * v32 + 64 * v64 + 128 * v128
        rows.append((k, n, v, v32, 32 * v32 + 64 * (v - v32), 32 * v32 + 128 * (v - v32), v64, v128, by_size))
    rows.sort(key=lambda r: -r[5] * r[1])
    with open(dst, "w") as f:
        f.write("# rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum (+ a pass with TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum)"
                " -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-rooflines --no-ragged-workload\n")
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "RDREQ_mean", "RDREQ_32B_mean", "read_bytes_if_64B_requests", "read_bytes_if_128B_requests",
                    "RDREQ_64B_mean", "RDREQ_128B_mean", "read_bytes_by_size"])
        fmt = lambda x: "" if x is None else f"{x:.0f}"
        for r in rows[:30]:
            w.writerow([r[0], r[1], fmt(r[2]), fmt(r[3]), fmt(r[4]), fmt(r[5]), fmt(r[6]), fmt(r[7]), fmt(r[8])])


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "rdreq":
        rdreq(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None)
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
