#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; rm -rf $O/pmc_g3
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS -d $O/pmc_g3 -- python $R/scratch/gemm_pmc.py > $O/pmc_g3.log 2>&1 || { tail -20 $O/pmc_g3.log; exit 1; }
cd $R
python - <<'PY'
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_g3/**/*counter_collection.csv",recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"]
    if "gemm" in k:
        k=k.split("(")[0].replace("void mmdti::","")
        agg[(k,r["Grid_Size"] if "Grid_Size" in r else "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()):
    d={c:sum(x)/len(x) for c,x in v.items()}
    wc=d.get("SQ_WAVE_CYCLES",1)
    print(k, "n=%d"%len(next(iter(v.values()))), "mfma_busy=%.0fM"%(d.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/1e6), "wait_any=%.0f%%"%(100*d.get("SQ_WAIT_ANY",0)/wc), "wait_inst=%.0f%%"%(100*d.get("SQ_WAIT_INST_ANY",0)/wc), "lds_wait=%.0f%%"%(100*d.get("SQ_WAIT_INST_LDS",0)/wc), "bank_conf=%.1fM"%(d.get("SQ_LDS_BANK_CONFLICT",0)/1e6), "lds_insts=%.1fM"%(d.get("SQ_INSTS_LDS",0)/1e6))
PY
