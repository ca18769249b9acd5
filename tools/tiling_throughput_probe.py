#!/usr/bin/env python3
"""Saturated-throughput effect of the decode-GEMM tiling / K-split choice (4 streams): the tuner ranks candidates by
isolated back-to-back time, which favours many small tiles; with several batches in flight fewer, larger tiles use
fewer CU-seconds and less L2 traffic.  Coordinate descent over the decode shapes on top of the autotuned table.
(The captured hipGraphs bake the tilings in: the graph cache is cleared after every override.)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openviic_amd import native
from openviic_amd.builders import build_model
from openviic_amd.config import model_config
from openviic_amd.instance import InstanceList
from openviic_amd.utils.synthetic import SyntheticVocab, synthetic_features, synthetic_state_dict

lib = native.load()
vocab = SyntheticVocab(10201, 20)
model = build_model(model_config("standard_transformer", device="cuda"), vocab).eval()
model.load_state_dict(synthetic_state_dict(model.state_dict()), strict=False)
items = InstanceList(); items.region_features = synthetic_features(256, 50, 2048).cuda()
NS = 4
streams = [torch.cuda.Stream() for _ in range(NS)]


def run(nstreams=NS, steps=32, warm=12):
    def step(i):
        with torch.cuda.stream(streams[i % nstreams]):
            model.beam_search(items, batch_size=256, beam_size=5)
    with torch.no_grad():
        for i in range(warm): step(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(steps): step(i)
        torch.cuda.synchronize()
    return 256 * steps / (time.perf_counter() - t0)


def code(t, split=1, st=0):
    return t | split << 8 | (st if split > 1 else 0) << 16


def show(c):
    return "t%d" % (c & 0xff) + ("/s%d:t%d" % ((c >> 8) & 0xff, c >> 16) if (c >> 8) & 0xff > 1 else "")


base = run()
shapes = {"o+q (1280x512x512)": (1280, 512, 1, 512), "ffn2 (1280x512x2048)": (1280, 512, 1, 2048),
          "qkv (1280x3x512x512)": (1280, 512, 3, 512), "ffn1 (1280x2048x512)": (1280, 2048, 1, 512),
          "vocab (1280x10201x512)": (1280, 10201, 1, 512)}
tuned = {k: lib.ovc_gemm_tuned_get(*v) for k, v in shapes.items()}
print("autotuned", {k: show(v) for k, v in tuned.items()}, "-> %.0f captions/s (%d streams), %.0f (1 stream)" % (base, NS, run(1)), flush=True)
small = [3, 4, 5, 6, 7, 8, 9, 10]
cands = {
    "o+q (1280x512x512)": [code(t) for t in (3, 6, 10)] + [code(6, s, st) for s in (2, 4) for st in (3, 4, 5, 7, 8, 9) if 512 // s % (64 if st >= 7 else 32) == 0],
    "ffn2 (1280x512x2048)": [code(10)] + [code(10, s, st) for s in (2, 4) for st in (1, 2, 3, 4, 5, 7, 8, 9)],
    "qkv (1280x3x512x512)": [code(t) for t in (1, 2, 3, 7, 11, 4, 8)],
    "ffn1 (1280x2048x512)": [code(t) for t in (0, 1, 2, 3, 4, 5, 7)],
    "vocab (1280x10201x512)": [code(t) for t in (0, 1, 2, 3, 4, 7)],
}
for name, options in cands.items():
    res = []
    for c in options:
        if lib.ovc_gemm_tuned_set(*shapes[name], c) != 0:
            continue
        lib.ovc_graph_cache_clear()
        res.append((c, run()))
    lib.ovc_gemm_tuned_set(*shapes[name], tuned[name]); lib.ovc_graph_cache_clear()
    res.append((tuned[name], run()))
    best = max(res, key=lambda r: r[1])
    lib.ovc_gemm_tuned_set(*shapes[name], best[0]); lib.ovc_graph_cache_clear()
    print(name, " ".join("%s:%.0f" % (show(c), v) for c, v in res), "-> keep", show(best[0]), flush=True)
print("final %.0f captions/s (%d streams), %.0f (1 stream)" % (run(NS, 64), NS, run(1, 24)))
