#!/bin/bash
# usage: scratch/resusage.sh file.hip [filter]   -- per-kernel VGPR / scratch / LDS / occupancy from the compiler remarks
f=$1; pat=${2:-.}
cd /root/repo/mm-dti_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -c $f -o /tmp/ru.o -Rpass-analysis=kernel-resource-usage 2>&1 | \
 awk '/Function Name:/{name=$0; sub(/.*Function Name: /,"",name); sub(/ \[.*/,"",name)} /VGPRs:/{v=$NF; sub(/.*VGPRs: /,"",$0); v=$1} /AGPRs:/{sub(/.*AGPRs: /,"",$0); a=$1} /ScratchSize/{sub(/.*: /,"",$0); sc=$1} /Occupancy/{sub(/.*: /,"",$0); oc=$1} /LDS Size/{sub(/.*: /,"",$0); print name, "vgpr="v, "agpr="a, "scratch="sc, "occ="oc, "lds="$1}' | c++filt | sed -E "s/\(unsigned short const.*\)/()/" | grep -E "$pat"
