for v in old new old new; do echo $v; MMDTI_HIP_LIB=$PWD/scratch/ab/lib_$v.so python scratch/gbf_bench.py 2>&1 | grep -v amdgpu.ids; done
