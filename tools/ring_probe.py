#!/usr/bin/env python3
"""files -> strings at B = 256 with the host side in worker processes: the shared page-locked ring (workers collate where the copy
engine reads) against the copier thread (a collated batch comes back through the loader's shared memory and is copied into pinned
buffers).    python tools/ring_probe.py [files=4096] [repeat=4]"""
import os, sys, tempfile, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import torch
    from openviic_amd.builders import build_model
    from openviic_amd.config import model_config
    from openviic_amd.data import predict_feature_files
    from openviic_amd.utils.synthetic import synthetic_state_dict
    from openviic_amd.vocab import WordVocab
    V, T, N, D = 10201, 20, 50, 2048
    files = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    repeat = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    vocab = WordVocab(["<pad>", "<bos>", "<eos>", "<unk>"] + ["w%d" % i for i in range(V - 4)], max_caption_length=T)
    model = build_model(model_config("standard_transformer", d_feature=D, device="cuda:0"), vocab).eval()
    model.load_state_dict(synthetic_state_dict(model.state_dict(), seed=1234, mode="reference_init"), strict=False)
    tmp = tempfile.mkdtemp(prefix="ovc_ring_")
    feats = np.random.default_rng(0).standard_normal((files, N, D)).astype(np.float32)
    paths = []
    for i in range(files):
        path = os.path.join(tmp, "%06d.npz" % i)
        np.savez(path, region_features=feats[i])
        paths.append(path)
    many = paths * repeat
    want = predict_feature_files(model, vocab, many[:2048], batch_size=256, workers=12)         # warm: tuning, graphs, the ring (its largest form)
    assert predict_feature_files(model, vocab, many[:2048], batch_size=256, workers=4, direct=False) == want
    configs = [(8, True, 4), (8, False, 4), (12, True, 4), (12, False, 4), (12, True, 2), (12, False, 2)] if "slots" not in sys.argv else [(8, True, 4), (12, True, 4), (14, True, 4), (8, True, 4), (12, True, 4), (14, True, 4)]
    for workers, direct, slots in configs:
        predict_feature_files(model, vocab, many[:256 * 2 * slots], batch_size=256, workers=workers, direct=direct, slots=slots)     # streams, workspaces, graphs
        t0 = time.perf_counter()
        got = predict_feature_files(model, vocab, many, batch_size=256, workers=workers, direct=direct, slots=slots)
        dt = time.perf_counter() - t0
        assert got[:2048] == want
        print("B = 256, %2d workers, %d decode streams, %s: %7.0f captions/s over %d images (start-up included)"
              % (workers, slots, "shared page-locked ring" if direct else "copier thread          ", len(got) / dt, len(got)), flush=True)
    for p in paths:
        os.remove(p)
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
