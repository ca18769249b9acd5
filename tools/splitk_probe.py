#!/usr/bin/env python3
"""What a cross-workgroup K split could buy for the M = 1280 decode GEMMs (runs on the GPU box).

A K split of S turns  C[M,N] = A[M,K] W[N,K]^T  into S independent products over K/S that write S partial
outputs.  Its workgroup count, per-workgroup loads and output traffic equal those of the plain GEMM
M x (S*N) x (K/S), which the existing kernel can run: time that for every tiling and compare with M x N x K.
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openviic_amd import native
from tools.gemm_bench import TILINGS, time_native

CASES = [(1280, 512, 512), (1280, 512, 2048), (1280, 1536, 512), (256, 512, 512)]


def best(lib, M, N, K):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.zeros(N, device="cuda")
    y = torch.empty(M, N, device="cuda")
    times = []
    for t in range(len(TILINGS)):
        lib.ovc_debug_force_gemm_tiling(t)
        times.append(time_native(lib, x, w, b, y))
    lib.ovc_debug_force_gemm_tiling(-1)
    i = min(range(len(times)), key=times.__getitem__)
    return times[i], TILINGS[i]


def main():
    lib = native.load()
    for M, N, K in CASES:
        base, bt = best(lib, M, N, K)
        line = "%dx%dx%d: plain %.1f us (%s)" % (M, N, K, base, bt)
        for S in (2, 4, 8):
            if K % (S * 64):
                continue
            t, tt = best(lib, M, N * S, K // S)
            line += " | S=%d: %.1f us (%s)" % (S, t, tt)
        print(line, flush=True)


if __name__ == "__main__":
    main()
