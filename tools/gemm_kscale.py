#!/usr/bin/env python3
"""Time of one GEMM tiling as a function of K at fixed M, N: slope = steady-state main-loop rate, intercept = fixed
cost per launch (launch + prologue + epilogue + tail).   python tools/gemm_kscale.py M N tiling[,tiling...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openviic_amd import native
from tools.gemm_bench import tilings, time_tiling

def main():
    M, N = int(sys.argv[1]), int(sys.argv[2])
    want = [int(t) for t in sys.argv[3].split(",")]
    lib = native.load()
    names = {t: label for t, label, _ in tilings(lib)}
    for t in want:
        pts = []
        for K in (256, 512, 1024, 2048, 4096):
            x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5
            b = torch.randn(N, device="cuda"); y = torch.empty(M, N, device="cuda")
            pts.append((K, time_tiling(lib, x, w, b, y, t, 1)))
        (k0, t0), (k1, t1) = pts[1], pts[-1]
        slope = (t1 - t0) / (k1 - k0)
        print("%-22s M=%d N=%d: %s | slope %.4f us/k -> %.1f TF steady, intercept %.1f us" % (
            names[t], M, N, " ".join("K=%d:%.1f" % p for p in pts), slope, 2.0 * M * N / slope / 1e6, t0 - slope * k0), flush=True)

if __name__ == "__main__":
    main()
