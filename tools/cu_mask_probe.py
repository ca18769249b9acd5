#!/usr/bin/env python3
"""Spatial partitioning probe: decode streams created with hipExtStreamCreateWithCUMask, each confined to a subset of
the 256 CUs, against the plain 4-stream mode of bench.py (every stream on every CU).

    python tools/cu_mask_probe.py [f32|f16x3]

Masks tried: 4 streams x 64 CUs (mask bits contiguous, and strided by 4), 2 x 128, and 8 streams x 32.
"""
import ctypes, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openviic_amd.builders import build_model
from openviic_amd.config import model_config
from openviic_amd.engine import CaptionEngine
from openviic_amd.instance import InstanceList
from openviic_amd.utils.synthetic import SyntheticVocab, synthetic_features, synthetic_state_dict


def masked_stream(hip, bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (w * 32 + b) in bits) for w in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def main():
    precision = sys.argv[1] if len(sys.argv) > 1 else "f32"
    vocab = SyntheticVocab(10201, 20)
    model = build_model(model_config("standard_transformer", d_feature=2048, device="cuda:0"), vocab).eval()
    model.load_state_dict(synthetic_state_dict(model.state_dict(), seed=1234, mode="reference_init"), strict=False)
    B = 256
    feats = synthetic_features(B, 50, 2048, seed=0).cuda()
    hip = ctypes.CDLL("libamdhip64.so")

    def decode():
        items = InstanceList()
        items.region_features = feats
        return model.beam_search(items, batch_size=B, beam_size=5, out_size=1)

    def run(streams, steps=24, objective=2):
        model._engine = CaptionEngine(model, tune_concurrency=objective, precision=precision)
        with torch.no_grad():
            for i in range(3 * len(streams)):
                with torch.cuda.stream(streams[i % len(streams)]):
                    decode()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                with torch.cuda.stream(streams[i % len(streams)]):
                    decode()
            torch.cuda.synchronize()
        model._engine.release()
        return B * steps / (time.perf_counter() - t0)

    plain = [torch.cuda.Stream() for _ in range(4)]
    print("plain 4 streams: %.0f captions/s" % run(plain), flush=True)
    print("plain 1 stream : %.0f captions/s" % run(plain[:1], objective=1), flush=True)
    cases = {
        "4 x 64 CUs, contiguous bits": [set(range(64 * i, 64 * i + 64)) for i in range(4)],
        "4 x 64 CUs, bits strided by 4": [set(range(i, 256, 4)) for i in range(4)],
        "2 x 128 CUs, contiguous": [set(range(128 * i, 128 * i + 128)) for i in range(2)],
        "2 x 128 CUs, strided by 2": [set(range(i, 256, 2)) for i in range(2)],
        "8 x 32 CUs, contiguous": [set(range(32 * i, 32 * i + 32)) for i in range(8)],
        "4 x 128 CUs, overlapping halves": [set(range(0, 128)), set(range(128, 256)), set(range(0, 128)), set(range(128, 256))],
        "1 x 64 CUs (one stream alone)": [set(range(64))],
    }
    for name, masks in cases.items():
        streams = [masked_stream(hip, m) for m in masks]
        print("%-34s: %.0f captions/s" % (name, run(streams, objective=1)), flush=True)
    print("plain 4 streams again: %.0f captions/s" % run(plain), flush=True)


if __name__ == "__main__":
    main()
