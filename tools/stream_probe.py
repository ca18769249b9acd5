#!/usr/bin/env python3
"""Throughput (captions/s) of the standard B=256 beam-5 workload vs number of streams and forced GEMM tiling."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openviic_amd import native
from openviic_amd.builders import build_model
from openviic_amd.config import model_config
from openviic_amd.engine import CaptionEngine
from openviic_amd.instance import InstanceList
from openviic_amd.utils.synthetic import SyntheticVocab, synthetic_features, synthetic_state_dict

lib = native.load()
vocab = SyntheticVocab(10201, 20)
model = build_model(model_config("standard_transformer", device="cuda"), vocab).eval()
model.load_state_dict(synthetic_state_dict(model.state_dict()), strict=False)
items = InstanceList(); items.region_features = synthetic_features(256, 50, 2048).cuda()

def run(nstreams, steps=24, warm=6):
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

configs = [(-1, "auto")] + [(int(a), "tiling %s" % a) for a in sys.argv[1:]]
for tiling, name in configs:
    CaptionEngine.autotune = tiling < 0
    lib.ovc_debug_force_gemm_tiling(tiling)
    print(name, " ".join("s%d=%.0f" % (n, run(n)) for n in (1, 2, 3, 4, 5, 6, 8)), flush=True)
