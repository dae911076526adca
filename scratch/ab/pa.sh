timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "pair_attn" 2>&1 | tail -2
for v in old new old new; do echo "== $v"; MMDTI_HIP_LIB=$PWD/scratch/ab/lib_$v.so python scratch/pa_bench.py 0.1 20 tiled 2>&1 | grep -v "amdgpu.ids"; done
