#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; rm -rf $O/pmc_gbf $O/pmc_gbf2
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $O/pmc_gbf -- python $R/scratch/gbf_bench2.py > $O/pmc_gbf.log 2>&1 || { tail -20 $O/pmc_gbf.log; exit 1; }
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR -d $O/pmc_gbf2 -- python $R/scratch/gbf_bench2.py > $O/pmc_gbf2.log 2>&1 || { tail -20 $O/pmc_gbf2.log; exit 1; }
cd $R
python - <<'PY'
import csv,glob,collections
for d in ("gpurun_out/pmc_gbf","gpurun_out/pmc_gbf2"):
    f=glob.glob(d+"/**/*counter_collection.csv",recursive=True)[0]
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "gbf_bias" in k:
            agg["bwd" if "bwd" in k else "fwd"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        print(k, {c: round(sum(x)/len(x)/1e6,1) for c,x in v.items()})
PY
