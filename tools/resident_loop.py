#!/usr/bin/env python3
"""Decode one resident batch in a loop on ONE stream (hipGraph replay): the unit the per-kernel traces of small batches are taken on.

    python tools/resident_loop.py [B=1] [repeats=30] [config=standard_transformer] [plain] [eos] [early]
        plain: no hipGraph;  eos: weights whose captions end around step 9 of 20 (utils/synthetic.py::eos_biased_state_dict);
        early: beam_search(..., early_exit=True) -- ovc_beam_search_early, stops issuing steps once every beam has ended
    rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/resident_loop.py 1 10
Prints ms per batch.
"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from openviic_amd.builders import build_model
from openviic_amd.config import model_config
from openviic_amd.instance import InstanceList
from openviic_amd.utils.synthetic import SyntheticVocab, synthetic_boxes, synthetic_features, synthetic_state_dict


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    variant = sys.argv[3] if len(sys.argv) > 3 else "standard_transformer"
    if "plain" in sys.argv[4:]:
        os.environ["OVC_GRAPH"] = "0"
    vocab = SyntheticVocab(10201, 20)
    model = build_model(model_config(variant, d_feature=2048, device="cuda:0"), vocab).eval()
    sd = synthetic_state_dict(model.state_dict(), seed=1234, mode="reference_init")
    flags = sys.argv[4:]
    if "eos" in flags:
        from openviic_amd.utils.synthetic import eos_biased_state_dict
        sd = eos_biased_state_dict(sd, model.state_dict(), ramp=float(os.environ.get("EOS_RAMP", "2.0")), gain=float(os.environ.get("EOS_GAIN", "3.0")))
    model.load_state_dict(sd, strict=False)
    early = "early" in flags
    items = InstanceList()
    items.region_features = synthetic_features(B, 50, 2048, seed=0).cuda()
    if variant == "object_relation_transformer":
        items.region_boxes = synthetic_boxes(B, 50, seed=0).cuda()
    with torch.no_grad():
        for _ in range(4):
            ids, _ = model.beam_search(items, batch_size=B, beam_size=5, early_exit=early)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            model.beam_search(items, batch_size=B, beam_size=5, early_exit=early)
        torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / reps
    ends = (ids == 2).float().argmax(-1)[(ids == 2).any(-1)]
    print("B=%d %s%s%s: %.3f ms per batch, %.3f ms per caption, %.1f captions/s; steps issued %d of 20; captions with <eos>: %d of %d, ending at step %.1f on average"
          % (B, variant, " eos-biased" if "eos" in flags else "", " early-exit" if early else "", ms, ms / B, B / ms * 1e3,
             model._engine.last_steps_run, int((ids == 2).any(-1).sum()), B, float(ends.float().mean()) if ends.numel() else -1))


if __name__ == "__main__":
    main()
