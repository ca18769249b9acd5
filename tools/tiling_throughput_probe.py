#!/usr/bin/env python3
"""Saturated-throughput effect of the decode-GEMM tiling choice (3 streams): latency-optimal tilings spread a
small GEMM over all CUs, larger tiles use fewer CU-seconds.  Tries overrides on top of the autotuned table."""
import sys, os, time, itertools
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

def run(nstreams=3, steps=24, warm=6):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    def step(i):
        with torch.cuda.stream(streams[i % nstreams]):
            model.beam_search(items, batch_size=256, beam_size=5)
    with torch.no_grad():
        for i in range(warm): step(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(steps): step(i)
        torch.cuda.synchronize()
    return 256 * steps / (time.perf_counter() - t0)

base = run()
shapes = {"oq": (1280, 512, 1, 512), "ffn2": (1280, 512, 1, 2048), "qkv": (1280, 512, 3, 512), "ffn1": (1280, 2048, 1, 512),
          "vocab": (1280, 10201, 1, 512)}
tuned = {k: lib.ovc_gemm_tuned_get(*v) & 0xff for k, v in shapes.items()}   # plain tiling (low byte of the code)
print("autotuned", tuned, "-> %.0f captions/s (1 stream %.0f)" % (base, run(1)), flush=True)
for name, cands in (("oq", [3, 7, 4, 8, 5, 9, 6, 10]), ("ffn2", [3, 7, 5, 9, 8, 10, 6]), ("qkv", [1, 3, 7, 11, 2]), ("ffn1", [1, 3, 5, 7, 2]),
                    ("vocab", [0, 1, 2, 3, 7])):
    res = []
    for t in cands:
        if lib.ovc_gemm_tuned_set(*shapes[name], t) != 0:
            continue
        res.append((t, run(3), run(1)))
    best = max(res, key=lambda r: r[1])
    lib.ovc_gemm_tuned_set(*shapes[name], best[0])          # keep the throughput-best and move on
    print(name, " ".join("t%d:%.0f/%.0f" % r for r in res), "-> keep", best[0], flush=True)
print("final %.0f captions/s (3 streams), %.0f (1 stream)" % (run(3, 48), run(1, 24)))
