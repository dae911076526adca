cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; rm -rf $O/blas
python $R/scratch/blas_ref.py 2>&1 | grep -v amdgpu.ids
rocprofv3 --output-format csv --kernel-trace --stats -d $O/blas -- python $R/scratch/blas_ref.py > $O/blas.log 2>&1
f=$(find $O/blas -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4 $f | cut -c1-200 | head -30
find $O/blas -type f ! -name "*kernel_stats.csv" -delete
