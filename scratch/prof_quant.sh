#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; rm -rf $O/prof_q
rocprofv3 --output-format csv --kernel-trace -d $O/prof_q -- python $R/scratch/gemm_quant.py > $O/prof_q.log 2>&1 || { tail -20 $O/prof_q.log; exit 1; }
cat $O/prof_q.log | tail -14
