#!/bin/bash
# compile csrc/sn_assign.hip alone and print the register / scratch report of one kernel (default: assign_screen5_kernel)
cd "$(dirname "$0")/../schemanet-pytorch_amd" || exit 1
K=${1:-assign_screen5_kernel}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I../include -Icsrc -Wall -Wno-unused-function \
  -c csrc/sn_assign.hip -o build/sn_assign.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A12 "Function Name.*$K" | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|VGPRs Spill" | sed 's/.*remark: *//'
