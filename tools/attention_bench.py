#!/usr/bin/env python3
"""ovc_attention at B = 256, 8 heads x 64: the register-resident instances (up to 128 keys) against the key-tiled ones (round 4).

    python tools/attention_bench.py
us per launch (back-to-back launches, torch events), TFLOP/s of the two contractions (4 nq nk d_k per image and head).
"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from openviic_amd import ops


def main():
    b, h, dk = 256, 8, 64
    print("nq   nk   slots kernel      us per launch   TFLOP/s   us per (image, head, 1000 query-key pairs)")
    for nq, nk, m in ((50, 50, 0), (50, 50, 40), (100, 100, 0), (128, 128, 0), (100, 100, 40), (129, 129, 0), (196, 196, 0), (256, 256, 0),
                      (50, 246, 0), (400, 400, 0)):
        q, k, v = (torch.randn(b, n, h * dk, device="cuda") for n in (nq, nk, nk))
        mask = torch.zeros(b, 1, 1, nk, dtype=torch.bool, device="cuda")
        memory = None
        if m:
            memory = (torch.randn(1, m, h * dk, device="cuda") / dk, torch.randn(1, m, h * dk, device="cuda") / m, math.sqrt(dk), math.sqrt(m))
        for _ in range(3):
            ops.attention(q, k, v, h, mask=mask, memory=memory)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            ops.attention(q, k, v, h, mask=mask, memory=memory)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        flops = 4.0 * b * h * nq * (nk + m) * dk
        kind = "registers" if nk + m <= 192 and nq <= 128 else ("key tiles" if nk + m > 128 else "LDS scores")
        print("%4d %4d %4d  %-10s %10.1f %12.1f %14.3f" % (nq, nk, m, kind, us, flops / us / 1e6, us / (b * h * nq * (nk + m) / 1e3)))


if __name__ == "__main__":
    main()
