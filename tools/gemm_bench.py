#!/usr/bin/env python3
"""Micro-benchmark of the fp32 MFMA GEMM on the shapes of the captioning path (runs on the GPU box).

For every (shape, K-order class, K split) the engine issues: time of each tiling of that class
(ovc_debug_linear_tiling, back-to-back launches issued from C) and of torch.addmm (rocBLAS / hipBLASLt) as a
same-hardware reference.  Random operands.   python tools/gemm_bench.py [decode|encoder|small|all] [split]
With "split": also the opt-in split-precision classes (bf16 planes, gemm_split.h) with their error against an fp64 product.
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openviic_amd import native

DECODE = [  # (M, N, K, kchains, ksplit, note)
    (1280, 512, 512, 4, 1, "dec cross-q"), (1280, 512, 512, 4, 2, "dec o-proj /2"), (1280, 1536, 512, 4, 1, "dec qkv"),
    (1280, 2048, 512, 4, 1, "dec ffn1"), (1280, 512, 2048, 4, 4, "dec ffn2 /4"), (1280, 512, 2048, 4, 2, "dec ffn2 /2"), (1280, 512, 2048, 4, 1, "dec ffn2 /1"),
    (1280, 10201, 512, 4, 1, "vocab"), (256, 512, 512, 4, 1, "t=0 proj"), (256, 10201, 512, 4, 1, "t=0 vocab"),
]
# the reference's own operating points (B = 1 and B = 8 at beam 5: 5 and 40 decode rows; base_trainer.py:75-80)
SMALL = [(rows, n, k, 4, ks, "%s rows=%d" % (name, rows)) for rows in [int(r) for r in os.environ.get("OVC_BENCH_ROWS", "5,40").split(",")]
         for n, k, ks, name in ((512, 512, 1, "cross-q"), (512, 512, 2, "o-proj /2"), (1536, 512, 1, "qkv"), (2048, 512, 1, "ffn1"),
                                (512, 2048, 4, "ffn2 /4"))]
ENCODER = [
    (12800, 512, 2048, 1, 1, "feature proj"), (12800, 1536, 512, 1, 1, "enc qkv"), (12800, 512, 512, 1, 1, "enc o"),
    (12800, 2048, 512, 1, 1, "enc ffn1"), (12800, 3072, 512, 1, 1, "cross kv"),
    (10201, 1280, 512, 1, 1, "vocab^T"),       # the fp32 engine's vocabulary product: logits^T = fc . x^T, one chain
]


def tilings(lib):
    out, t = [], 0
    while lib.ovc_profile_kernel_name(t):
        name = lib.ovc_profile_kernel_name(t).decode()
        v = [int(x) for x in name[name.index("<") + 1:-1].split(",")]
        if "split_mfma" in name:
            out.append((t, "%dx%d b%d p%d" % (v[0], v[1], v[4], v[5]), 100 + v[5]))
        elif "rows16" in name:                      # gemm_rows16.h: 16-row tiles, the four-chain class, up to 112 rows
            out.append((t, "16x%d rows16" % (16 * v[0]), 4))
        else:
            out.append((t, "%dx%d w%d b%d c%d" % (v[0], v[1], v[4], v[5], v[6]), v[4] * v[6]))
        t += 1
    return out


def time_tiling(lib, x, w, b, y, t, ksplit, iters=40):
    M, K = x.shape
    N = w.shape[0]
    args = (x.data_ptr(), K, w.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, t, ksplit)
    if lib.ovc_debug_linear_tiling(*args, 5, native.stream_handle()) != 0:
        return None
    torch.cuda.synchronize()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    lib.ovc_debug_linear_tiling(*args, iters, native.stream_handle())
    stop.record()
    torch.cuda.synchronize()
    return start.elapsed_time(stop) / iters * 1e3


def time_planes(lib, x, w, planes, b, y, t, ksplit, iters=40):
    M, K = x.shape
    N = w.shape[0]
    args = (x.data_ptr(), K, w.data_ptr(), planes.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, t, ksplit)
    if lib.ovc_debug_linear_planes(*args, 5, native.stream_handle()) != 0:
        return None
    torch.cuda.synchronize()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    lib.ovc_debug_linear_planes(*args, iters, native.stream_handle())
    stop.record()
    torch.cuda.synchronize()
    return start.elapsed_time(stop) / iters * 1e3


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    split = "split" in sys.argv[2:]
    lib = native.load()
    shapes = ((DECODE if which in ("decode", "all") else []) + (ENCODER if which in ("encoder", "all") else []) +
              (SMALL if which == "small" else []))
    tl = tilings(lib)
    for M, N, K, chains, ksplit, note in shapes:
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") / K ** 0.5
        b = torch.randn(N, device="cuda")
        y = torch.empty(max(ksplit, 1), M, N, device="cuda")
        for _ in range(5):
            torch.addmm(b, x, w.t())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            torch.addmm(b, x, w.t())
        e1.record()
        torch.cuda.synchronize()
        ref = e0.elapsed_time(e1) / 30 * 1e3
        flops = 2.0 * M * N * K
        cells = []
        for t, label, c in tl:
            if c != chains:
                continue
            us = time_tiling(lib, x, w, b, y, t, ksplit)
            cells.append((us if us is not None else 1e9, label))
        cells.sort()
        print("%-16s %5dx%5dx%4d c%d /%d | torch %6.1f us | ideal %5.1f | %s" % (
            note, M, N, K, chains, ksplit, ref, flops / 157.3e6,
            "  ".join("%s: %.1f (%.0f TF)" % (l, u, flops / u / 1e6) for u, l in cells if u < 1e8)), flush=True)
        if not split:
            continue
        exact = (x.double() @ w.double().t())
        scale = exact.abs().mean().item()
        for cls in (chains, 103, 104):
            cells, err = [], None
            for t, label, c in tl:
                if c != cls:
                    continue
                us = time_tiling(lib, x, w, b, y, t, ksplit)
                if us is None:
                    continue
                cells.append((us, label))
                if err is None:      # (ksplit > 1: raw partial products, no bias)
                    want = exact + b.double() if ksplit <= 1 else exact
                    err = (y[:max(ksplit, 1)].double().sum(0) - want).abs().max().item() / scale
            cells.sort()
            if cls > 100:       # the same tilings on pre-cut weights (ovc_split_weight): bit-identical, W bypasses LDS
                mode = cls - 100
                planes = torch.empty(lib.ovc_split_weight_bytes(N, K, mode), dtype=torch.uint8, device="cuda")
                assert lib.ovc_split_weight(w.data_ptr(), N, K, mode, planes.data_ptr(), native.stream_handle()) == 0
                direct, same = [], True
                for t, label, c in tl:
                    if c != cls:
                        continue
                    y.zero_()
                    us = time_tiling(lib, x, w, b, y, t, ksplit)
                    ref_y = y.clone()
                    us2 = time_planes(lib, x, w, planes, b, y, t, ksplit)
                    if us is None or us2 is None:
                        continue
                    same = same and torch.equal(ref_y, y)
                    direct.append((us2, label))
                direct.sort()
                print("      planes:  bit-identical %s | %s" % (same, "  ".join("%s: %.1f (%.0f TF-eq)" % (l, u, flops / u / 1e6) for u, l in direct)), flush=True)
            print("    class %3d: max |err| / mean |y| = %.2e | %s" % (
                cls, err, "  ".join("%s: %.1f (%.0f TF-eq)" % (l, u, flops / u / 1e6) for u, l in cells)), flush=True)


if __name__ == "__main__":
    main()
