#!/usr/bin/env python3
"""Micro-benchmark of the fp32 MFMA GEMM on the shapes of the captioning path (runs on the GPU box).

For every shape: time of each tiling (forced through ovc_debug_force_gemm_tiling), of the automatic
choice, and of torch.mm (rocBLAS / hipBLASLt) as a same-hardware reference.  Random operands.
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openviic_amd import native, ops

SHAPES = [  # (M, N, K, note)
    (1280, 512, 512, "dec o/q proj"), (1280, 1536, 512, "dec qkv (3 seg)"), (1280, 2048, 512, "dec ffn1"),
    (1280, 512, 2048, "dec ffn2"), (1280, 10201, 512, "vocab"), (256, 512, 512, "t=0 proj"),
    (256, 10201, 512, "t=0 vocab"), (12800, 512, 2048, "feature proj"), (12800, 1536, 512, "enc qkv"),
    (12800, 512, 512, "enc o"), (12800, 2048, 512, "enc ffn1"), (12800, 512, 2048, "enc ffn2"),
    (12800, 3072, 512, "cross kv (6 seg)"),
]
TILINGS = ["128x128", "64x128", "128x64", "64x64", "32x64k2", "64x32k2", "32x32k4", "64x64b64", "32x64k2b", "64x32k2b", "32x32k4b", "64x128b"]


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        fn()
    stop.record()
    torch.cuda.synchronize()
    return start.elapsed_time(stop) / iters * 1e3       # microseconds


def time_native(lib, x, w, b, y, iters=50):
    """Back-to-back launches issued from C: no Python between kernels (host launch rate still applies)."""
    M, K = x.shape
    N = w.shape[0]
    args = (x.data_ptr(), K, w.data_ptr(), b.data_ptr(), y.data_ptr(), M, N)
    lib.ovc_debug_repeat_linear(*args, 5, native.stream_handle())
    torch.cuda.synchronize()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    lib.ovc_debug_repeat_linear(*args, iters, native.stream_handle())
    stop.record()
    torch.cuda.synchronize()
    return start.elapsed_time(stop) / iters * 1e3


def main():
    lib = native.load()
    dev = "cuda"
    print("%-22s %8s | %s | %8s %8s | TF(auto) TF(best) ideal_us" % ("shape", "note", " ".join("%8s" % t for t in TILINGS), "auto", "torch"))
    for M, N, K, note in SHAPES:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        times = []
        y = torch.empty(M, N, device=dev)
        ref_y = torch.addmm(b, x, w.t())
        for t in range(len(TILINGS)):
            lib.ovc_debug_force_gemm_tiling(t)
            times.append(time_native(lib, x, w, b, y))
            err = (y - ref_y).abs().max().item()
            if err > 2e-3:
                print("  !! tiling %s differs from torch.addmm by %.3e" % (TILINGS[t], err))
        lib.ovc_debug_force_gemm_tiling(-1)
        auto = time_native(lib, x, w, b, y)
        ref = timeit(lambda: torch.addmm(b, x, w.t()))
        flops = 2.0 * M * N * K
        print("%-22s %8s | %s | %8.1f %8.1f | %7.1f %7.1f %7.1f" % (
            "%dx%dx%d" % (M, N, K), note[:8], " ".join("%8.1f" % t for t in times), auto, ref,
            flops / auto / 1e6, flops / min(times) / 1e6, flops / 157.3e6))


if __name__ == "__main__":
    main()
