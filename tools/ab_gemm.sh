#!/bin/bash
# same-box A/B of two builds of the library on the GEMM micro-benchmarks: tools/libovc_base.bin vs the in-tree build
for L in tools/libovc_base.bin openviic_amd/csrc/libovc.so; do
  echo "=== $L"
  OVC_LIBRARY=$PWD/$L python tools/gemm_kscale.py 1280 2048 9,7 2>&1 | grep -v amdgpu
  OVC_LIBRARY=$PWD/$L python tools/gemm_kscale.py 1280 512 14,9 2>&1 | grep -v amdgpu
  OVC_LIBRARY=$PWD/$L python tools/gemm_kscale.py 12800 1536 0,3 2>&1 | grep -v amdgpu
  OVC_LIBRARY=$PWD/$L python tools/gemm_kscale.py 1280 10201 7,9 2>&1 | grep -v amdgpu
done
