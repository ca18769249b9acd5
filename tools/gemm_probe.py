#!/usr/bin/env python3
"""Per-workgroup efficiency probe of the 128x128 / 64x128 / 64x64 tilings: shapes that give exactly 1, 2, 4
workgroups per CU (no tail), K = 512 and 2048."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openviic_amd import native
from gemm_bench import time_native, TILINGS

lib = native.load()
for (M, N, K) in [(2048, 2048, 512), (4096, 2048, 512), (8192, 2048, 512), (4096, 2048, 2048), (8192, 4096, 2048),
                  (1024, 1024, 512), (2048, 1024, 512), (2048, 2048, 2048)]:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda"); y = torch.empty(M, N, device="cuda")
    row = []
    for t in (0, 1, 3, 7, 11):
        lib.ovc_debug_force_gemm_tiling(t)
        us = time_native(lib, x, w, b, y, iters=30)
        row.append("%s %6.1fus %5.1fTF" % (TILINGS[t], us, 2.0 * M * N * K / us / 1e6))
    lib.ovc_debug_force_gemm_tiling(-1)
    print("%5dx%5dx%5d | " % (M, N, K) + " | ".join(row))
