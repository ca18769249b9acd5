#!/usr/bin/env python3
"""The reference's test loop at ITS batch size (trainers/base_trainer.py:75-80: batch_size = 1), files -> strings on the GPU box:
one image per beam search, fed by one thread or by DataLoader workers, with and without early exit.

    python tools/b1_loop_probe.py [images=1024]
"""
import os, sys, tempfile, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from openviic_amd.builders import build_model
from openviic_amd.config import model_config
from openviic_amd.data import predict_feature_files
from openviic_amd.utils.synthetic import eos_biased_state_dict, synthetic_state_dict
from openviic_amd.vocab import WordVocab


def main():
    V, T, N, D = 10201, 20, 50, 2048
    images = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    vocab = WordVocab(["<pad>", "<bos>", "<eos>", "<unk>"] + ["w%d" % i for i in range(V - 4)], max_caption_length=T)
    model = build_model(model_config("standard_transformer", d_feature=D, device="cuda:0"), vocab).eval()
    sd = synthetic_state_dict(model.state_dict(), seed=1234, mode="reference_init")
    tmp = tempfile.mkdtemp(prefix="ovc_b1_")
    feats = np.random.default_rng(0).standard_normal((images, N, D)).astype(np.float32)
    paths = []
    for i in range(images):
        path = os.path.join(tmp, "%06d.npz" % i)
        np.savez(path, region_features=feats[i])
        paths.append(path)
    for label, weights in (("random-init weights (no caption ends)", sd), ("eos-biased weights (captions end around step 9)", eos_biased_state_dict(sd, model.state_dict()))):
        model.load_state_dict(weights, strict=False)
        print(label)
        predict_feature_files(model, vocab, paths[:256], batch_size=1, beam_size=5, workers=4, slots=8)    # streams, workspaces, graphs, tuning
        want = None
        for workers, early, slots in ((0, False, 4), (8, False, 1), (8, False, 2), (8, False, 4), (8, False, 6), (12, False, 4), (8, True, 4)):
            t0 = time.perf_counter()
            got = predict_feature_files(model, vocab, paths, batch_size=1, beam_size=5, workers=workers, early_exit=early, slots=slots)
            dt = time.perf_counter() - t0
            want = want or got
            assert got == want
            print("  B = 1, workers %2d, decode streams %d, early_exit %-5s: %7.1f captions/s, %.2f ms per caption (start-up included)"
                  % (workers, slots, early, len(got) / dt, 1e3 * dt / len(got)), flush=True)
    for p in paths:
        os.remove(p)
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
