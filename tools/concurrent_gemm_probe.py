#!/usr/bin/env python3
"""Aggregate TFLOP/s of the decode-layer GEMM sequence when S independent streams run it concurrently."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openviic_amd import native
lib = native.load()
SEQ = [(1280, 1536, 512), (1280, 512, 512), (1280, 512, 512), (1280, 512, 512), (1280, 2048, 512), (1280, 512, 2048)]
BIG = [(12800, 2048, 512), (12800, 512, 2048), (12800, 1536, 512)]
VOC = [(1280, 10201, 512)]

def bufs(shapes):
    out = []
    for M, N, K in shapes:
        out.append((torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") / K ** 0.5,
                    torch.randn(N, device="cuda"), torch.empty(M, N, device="cuda"), M, N, K))
    return out

def run(shapes, S, reps):
    streams = [torch.cuda.Stream() for _ in range(S)]
    sets = [bufs(shapes) for _ in range(S)]
    def once():
        for r in range(reps):
            for s in range(S):
                with torch.cuda.stream(streams[s]):
                    h = native.stream_handle()
                    for x, w, b, y, M, N, K in sets[s]:
                        lib.ovc_debug_repeat_linear(x.data_ptr(), K, w.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, 1, h)
    once(); torch.cuda.synchronize()
    t0 = time.perf_counter(); once(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    flops = S * reps * sum(2.0 * M * N * K for M, N, K in shapes)
    return flops / dt / 1e12

for name, shapes, reps in (("decode layer", SEQ, 60), ("vocab", VOC, 60), ("encoder big", BIG, 20)):
    print(name, " ".join("S%d=%.1fTF" % (S, run(shapes, S, reps)) for S in (1, 2, 3, 4, 6)), flush=True)
