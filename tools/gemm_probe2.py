import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openviic_amd import native
from gemm_bench import time_native, TILINGS
lib = native.load()
for (M, N, K) in [(8192, 8192, 512), (8192, 8192, 2048)]:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda"); y = torch.empty(M, N, device="cuda")
    for t in (0, 1, 3):
        lib.ovc_debug_force_gemm_tiling(t)
        for rep in range(3):
            us = time_native(lib, x, w, b, y, iters=8)
        print("%dx%dx%d %s %.1fus %.1fTF" % (M, N, K, TILINGS[t], us, 2.0 * M * N * K / us / 1e6))
