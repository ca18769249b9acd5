#!/usr/bin/env python3
"""Where does a DataLoader over feature files spend its time?  (host only; run on the GPU box for its cores and file system)

    python tools/loader_probe.py [images=2048] [dir=/tmp]

Prints, for the {image}.npz files of the BASELINE shape (50 x 2048 fp32, 0.41 MB): the file system, one-process parse rates
(np.load / zipfile / raw read), and batches per second out of `feature_file_loader` for several worker counts with and without
the pinning thread -- no GPU work, so that the loader's own ceiling is visible.
"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from openviic_amd.data import batch_from_feature_files, feature_file_loader, load_feature_file


def main():
    images = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    root = sys.argv[2] if len(sys.argv) > 2 else tempfile.gettempdir()
    tmp = tempfile.mkdtemp(prefix="ovc_loader_", dir=root)
    os.system("df -hT %s | tail -1; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null" % tmp)
    feats = np.random.default_rng(0).standard_normal((images, 50, 2048)).astype(np.float32)
    paths = []
    for i in range(images):
        path = os.path.join(tmp, "%06d.npz" % i)
        np.savez(path, region_features=feats[i])
        paths.append(path)
    t0 = time.perf_counter()
    for p in paths[:512]:
        open(p, "rb").read()
    raw = 512 / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    for p in paths[:512]:
        load_feature_file(p)
    parse = 512 / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    for i in range(0, 1024, 256):
        batch_from_feature_files(paths[i:i + 256])
    collate = 1024 / (time.perf_counter() - t0)
    print("one process: raw read %.0f files/s, np.load %.0f files/s, load + collate %.0f images/s" % (raw, parse, collate), flush=True)
    many = paths * max(1, 8192 // len(paths))
    for pin in (False, True):
        for workers in (2, 4, 8, 12):
            t0 = time.perf_counter()
            n = 0
            first = None
            for fields in feature_file_loader(many, 256, workers, pin_memory=pin):
                n += fields["region_features"].shape[0]
                if first is None:
                    first = time.perf_counter() - t0
            dt = time.perf_counter() - t0
            print("loader only: workers %2d pin_memory %-5s %7.0f images/s (first batch after %.2f s; steady %.0f images/s)"
                  % (workers, pin, n / dt, first, (n - 256) / max(dt - first, 1e-9)), flush=True)
    if len(sys.argv) > 3 and sys.argv[3] == "gpu":
        # the whole prediction loop under cProfile: where does the CONSUMER (the launching thread) spend its time?
        import cProfile
        import pstats
        from openviic_amd.builders import build_model
        from openviic_amd.config import model_config
        from openviic_amd.data import predict_feature_files
        from openviic_amd.utils.synthetic import synthetic_state_dict
        from openviic_amd.vocab import WordVocab
        V, T = 10201, 20
        vocab = WordVocab(["<pad>", "<bos>", "<eos>", "<unk>"] + ["w%d" % i for i in range(V - 4)], max_caption_length=T)
        model = build_model(model_config("standard_transformer", d_feature=2048, device="cuda:0"), vocab).eval()
        model.load_state_dict(synthetic_state_dict(model.state_dict(), seed=1234, mode="reference_init"), strict=False)
        predict_feature_files(model, vocab, many[:1024], batch_size=256, slots=2, workers=4)
        predict_feature_files(model, vocab, many[:2560], batch_size=256, slots=2, workers=4)
        os.environ["OVC_PREDICT_TRACE"] = "1"
        predict_feature_files(model, vocab, many[:1536], batch_size=256, slots=2, workers=8)
        os.environ["OVC_PREDICT_TRACE"] = "0"
        for workers, context in ((4, "forkserver"), (8, "forkserver"), (12, "forkserver"), (14, "forkserver"), (8, "spawn")):
            t0 = time.perf_counter()
            got = predict_feature_files(model, vocab, many * 4, batch_size=256, slots=2, workers=workers, loader_context=context)
            print("predict_feature_files, %d %s workers: %.0f captions/s over %d images (start-up included)"
                  % (workers, context or "fork", len(got) / (time.perf_counter() - t0), len(got)), flush=True)
        for workers in ():
            prof = cProfile.Profile()
            t0 = time.perf_counter()
            prof.enable()
            predict_feature_files(model, vocab, many, batch_size=256, slots=2, workers=workers)
            prof.disable()
            print("predict_feature_files, %d workers: %.0f captions/s" % (workers, len(many) / (time.perf_counter() - t0)), flush=True)
            pstats.Stats(prof).sort_stats("tottime").print_stats(14)
    for p in paths:
        os.remove(p)
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
