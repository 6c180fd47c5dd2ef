#!/bin/bash
# compile csrc/sn_assign.hip alone with the resource-usage remarks of one kernel ($1, default assign_screen4), and its ISA to /tmp/s4.s
K=${1:-assign_screen4}
cd /root/repo/schemanet-pytorch_amd || exit 1
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I../include -Icsrc -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS -Rpass-analysis=kernel-resource-usage -c csrc/sn_assign.hip -o build/sn_assign.o 2>&1 | grep -E "error|$K" -A6 | grep -E 'error|Function Name|VGPRs:|AGPRs|Spill|ScratchSize'
/opt/rocm/bin/hipcc $FLAGS -S --cuda-device-only -o /tmp/sn_assign.s csrc/sn_assign.hip 2>/dev/null
