#!/usr/bin/env python3
"""Tilings chosen by ovc_gemm_tune for the decode shapes under objective 1 (isolated) and 2/4/8 co-running copies."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openviic_amd import native
lib = native.load()
shapes = [(1280, 512, 1, 512), (1280, 512, 1, 2048), (1280, 512, 3, 512), (1280, 2048, 1, 512), (1280, 10201, 1, 512),
          (12800, 512, 1, 2048), (12800, 512, 3, 512), (12800, 2048, 1, 512), (12800, 512, 1, 512)]
scratch = torch.empty(120 << 20, dtype=torch.float32, device="cuda").normal_()
def show(c):
    return "t%d" % (c & 0xff) + ("/s%d:t%d" % ((c >> 8) & 0xff, c >> 16) if (c >> 8) & 0xff > 1 else "")
for obj in (1, 2, 4, 8):
    lib.ovc_gemm_tune_objective(obj)
    out = []
    for i, s in enumerate(shapes):
        key = (s[0] + obj, s[1], s[2], s[3])            # a distinct M per objective: the table is keyed by shape
        assert lib.ovc_gemm_tune(*key, scratch.data_ptr(), scratch.numel() * 4, native.stream_handle()) == 0
        out.append(show(lib.ovc_gemm_tuned_get(*key)))
    print("objective", obj, out, flush=True)
