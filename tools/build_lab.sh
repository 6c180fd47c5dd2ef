#!/bin/bash
# a lab build of the library with -D flags on ONE source file: tools/build_lab.sh NAME sn_assign "-DSN_S1_SADDR=0" -> tools/lab/bin/NAME.so
# (A/B inside one gpurun call: SN_LIB_PATH=tools/lab/bin/NAME.so python tools/<script>.py)
set -e
cd "$(dirname "$0")/../schemanet-pytorch_amd"
name=$1; src=$2; shift 2
mkdir -p ../tools/lab/bin build/lab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I../include -Icsrc -Wall -Wno-unused-function "$@" -c csrc/$src.hip -o build/lab/${src}_$name.o
objs=$(ls build/*.o | grep -v "build/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../tools/lab/bin/$name.so $objs build/lab/${src}_$name.o
echo built tools/lab/bin/$name.so
