#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; rm -rf $O/pmc_g1 $O/pmc_g2
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $O/pmc_g1 -- python $R/scratch/gemm_pmc2.py > $O/pmc_g1.log 2>&1 || { tail -20 $O/pmc_g1.log; exit 1; }
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT -d $O/pmc_g2 -- python $R/scratch/gemm_pmc2.py > $O/pmc_g2.log 2>&1 || { tail -20 $O/pmc_g2.log; exit 1; }
cd $R
python - <<'PY'
import csv,glob,collections
for d in ("gpurun_out/pmc_g1","gpurun_out/pmc_g2"):
    f=[p for p in glob.glob(d+"/**/*counter_collection.csv",recursive=True)]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "gemm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(d[-6:],k,sum(v)/len(v))
PY
