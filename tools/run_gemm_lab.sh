#!/bin/bash
# every lab build of tools/lab/bin against the shipped library, C2 class products: tools/run_gemm_lab.sh OUT.log
out=$1; : > $out
python tools/time_gemm_tile.py c2 >> $out 2>&1
for f in tools/lab/bin/*.so; do SN_LIB_PATH=$f python tools/time_gemm_tile.py c2 >> $out 2>&1; done
grep -v amdgpu.ids $out | sed 's/span [0-9]* cycles, //' | cut -c1-200
