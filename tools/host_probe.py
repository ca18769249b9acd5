#!/usr/bin/env python3
"""How long does the host take to ENQUEUE one beam_search batch (all launches are asynchronous)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openviic_amd.builders import build_model
from openviic_amd.config import model_config
from openviic_amd.instance import InstanceList
from openviic_amd.utils.synthetic import SyntheticVocab, synthetic_features, synthetic_state_dict

vocab = SyntheticVocab(10201, 20)
model = build_model(model_config("standard_transformer", device="cuda"), vocab).eval()
model.load_state_dict(synthetic_state_dict(model.state_dict()), strict=False)
items = InstanceList(); items.region_features = synthetic_features(256, 50, 2048).cuda()
stream = torch.cuda.Stream()          # the default (null) stream cannot be captured into a hipGraph
with torch.no_grad(), torch.cuda.stream(stream):
    for _ in range(3): model.beam_search(items, batch_size=256, beam_size=5)
    torch.cuda.synchronize()
    for rep in range(4):
        t0 = time.perf_counter()
        model.beam_search(items, batch_size=256, beam_size=5)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("enqueue %.2f ms, then wait %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
