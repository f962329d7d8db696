#!/bin/bash
# register / spill / LDS usage of the kernels of one csrc file:   scripts/kernel_regs.sh gemm.hip [name-filter] [-DFLAGS...]
f=$1; pat=${2:-.}; shift; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c dexbotic_amd/csrc/$f -o /tmp/_kr.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|    VGPRs:|AGPRs:|VGPRs Spill|ScratchSize" \
 | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | paste - - - - - | grep -E "$pat" | sed -e 's/Function Name: //' | c++filt | cut -c1-220
