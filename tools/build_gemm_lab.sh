#!/bin/bash
# lab builds of the library with -D flags on csrc/sn_gcn.hip only: tools/build_gemm_lab.sh NAME "-DSN_GEMM_ABLATE=1" -> tools/lab/bin/NAME.so
# (A/B inside one gpurun call: SN_LIB_PATH=tools/lab/bin/NAME.so python tools/<script>.py)
set -e
cd "$(dirname "$0")/../schemanet-pytorch_amd"
name=$1; shift
mkdir -p ../tools/lab/bin build/lab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I../include -Icsrc -Wall -Wno-unused-function "$@" -c csrc/sn_gcn.hip -o build/lab/sn_gcn_$name.o
objs=$(ls build/*.o | grep -v sn_gcn.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../tools/lab/bin/$name.so $objs build/lab/sn_gcn_$name.o
echo built tools/lab/bin/$name.so
