#!/usr/bin/env python3
"""Which kernels does torch.addmm (hipBLASLt / rocBLAS fp32) pick for the decode shapes?  Run under rocprofv3 --kernel-trace."""
import torch
for M, N, K in [(1280, 1536, 512), (1280, 512, 2048), (1280, 512, 512), (1280, 2048, 512), (1280, 10201, 512)]:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
    for _ in range(3):
        torch.addmm(b, x, w.t())
torch.cuda.synchronize()
