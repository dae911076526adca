#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; rm -rf $O/pmc_at $O/pmc_at2
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $O/pmc_at -- python $R/scratch/attn_bench.py 0.1 3 > $O/pmc_at.log 2>&1 || { tail -20 $O/pmc_at.log; exit 1; }
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE -d $O/pmc_at2 -- python $R/scratch/attn_bench.py 0.1 3 > $O/pmc_at2.log 2>&1 || { tail -20 $O/pmc_at2.log; exit 1; }
python3 - <<PY
import csv, glob, collections
for d in ("$O/pmc_at", "$O/pmc_at2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].split("(")[0].replace("void mmdti::", "")
            if "attn" not in n: continue
            acc[n][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(n, r["Counter_Name"])] += 1
    for n in acc:
        print(n, {c: round(v / max(1, cnt[(n, c)]) / 1e6, 2) for c, v in acc[n].items()}, "(millions per dispatch-row)")
PY
