#!/usr/bin/env python3
"""files -> captions at the BASELINE shapes, and the reference loop's own operating points (runs on the GPU box).

    python tools/predict_probe.py [images=1024]

1. resident-feature latency at the batch sizes the reference's loaders use: B = 1 (test loader,
   trainers/base_trainer.py:75-80) and B = 10 (DICT_BATCH_SIZE // beam-style validation batches), one stream, hipGraph
   replay: ms per batch and per caption;
2. {image}.npz feature files on local disk (50 x 2048 fp32 regions, 400 KB each) -> captions: the plain sequential loop
   (load, collate, .to(device), beam_search, strings) against openviic_amd.data.predict_feature_files (
   pinned staging buffers, copy stream, 2 decode streams, one host thread) at B = 256; and both at B = 1;
3. the same loop fed by DataLoader worker processes (predict_feature_files(workers=N)): forkserver / spawn / fork workers.
Prints one JSON line.
"""
import json, os, sys, tempfile, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from openviic_amd.builders import build_model
from openviic_amd.config import model_config
from openviic_amd.data import batch_from_feature_files, predict_feature_files
from openviic_amd.instance import InstanceList
from openviic_amd.utils.synthetic import SyntheticVocab, synthetic_features, synthetic_state_dict
from openviic_amd.vocab import WordVocab, captions_from_ids

def main():
    V, T, N, D = 10201, 20, 50, 2048
    images = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    words = ["<pad>", "<bos>", "<eos>", "<unk>"] + ["w%d" % i for i in range(V - 4)]
    vocab = WordVocab(words, max_caption_length=T)
    cfg = model_config("standard_transformer", d_feature=D, device="cuda:0")
    model = build_model(cfg, vocab).eval()
    model.load_state_dict(synthetic_state_dict(model.state_dict(), seed=1234, mode="reference_init"), strict=False)
    out = {}

    with torch.no_grad():
        for B in (1, 10, 32):
            feats = synthetic_features(B, N, D, seed=0).cuda()
            items = InstanceList(); items.region_features = feats
            for _ in range(4):
                model.beam_search(items, batch_size=B, beam_size=5)
            torch.cuda.synchronize()
            reps = 30
            t0 = time.perf_counter()
            for _ in range(reps):
                model.beam_search(items, batch_size=B, beam_size=5)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / reps
            out["resident_B%d" % B] = {"ms_per_batch": round(ms, 3), "ms_per_caption": round(ms / B, 3), "captions_per_s": round(B / ms * 1e3, 1)}
            print("[probe] resident B=%d: %.3f ms per batch" % (B, ms), file=sys.stderr, flush=True)

        tmp = tempfile.mkdtemp(prefix="ovc_feats_")
        feats = synthetic_features(images, N, D, seed=1).numpy()
        paths = []
        for i in range(images):
            path = os.path.join(tmp, "%06d.npz" % i)
            np.savez(path, region_features=feats[i])
            paths.append(path)
        def sequential(batch_size, subset):
            res = []
            for i in range(0, len(subset), batch_size):
                items = batch_from_feature_files(subset[i:i + batch_size], device="cuda")
                outs, _ = model.beam_search(items, batch_size=items.batch_size, beam_size=5, out_size=1)
                res += list(zip(items.filename, captions_from_ids(vocab, outs)))
            return res
        for B, subset in ((256, paths), (1, paths[:512])):
            sequential(B, subset[:2 * B]); predict_feature_files(model, vocab, subset[:6 * B], batch_size=B, slots=2)   # warm: tuning, graphs
            torch.cuda.synchronize()
            t0 = time.perf_counter(); a = sequential(B, subset); torch.cuda.synchronize(); ts = time.perf_counter() - t0
            t0 = time.perf_counter(); b = predict_feature_files(model, vocab, subset, batch_size=B, slots=2); tp = time.perf_counter() - t0
            assert a == b
            t0 = time.perf_counter()
            for i in range(0, len(subset), B):
                batch_from_feature_files(subset[i:i + B])
            tl = time.perf_counter() - t0
            out["files_B%d" % B] = {"images": len(subset), "sequential_captions_per_s": round(len(subset) / ts, 1),
                                    "pipelined_captions_per_s": round(len(subset) / tp, 1),
                                        "host_load_and_collate_only_captions_per_s": round(len(subset) / tl, 1),
                                    "feature_MB_per_image": round(N * D * 4 / 1e6, 3)}
            print("[probe] files B=%d: sequential %.1f, pipelined %.1f, host loading alone %.1f captions/s"
                  % (B, len(subset) / ts, len(subset) / tp, len(subset) / tl), file=sys.stderr, flush=True)
        # The host side in DataLoader worker processes (round 4): batches arrive through shared memory and are staged into a ring
        # of pinned buffers by a copier thread.  Whole call including worker start-up, over the file set repeated (page cache
        # warm: parsing + collating is the bound); every run's first 2048 strings are checked against the one-thread loop's.
        B = 256
        many = paths * max(1, 16384 // len(paths))
        cores = len(os.sched_getaffinity(0))
        base = dict(out["files_B256"])
        out["files_B256_loader"] = {"images": len(many), "usable_cores": cores}
        want = None
        many = many * 2
        out["files_B256_loader"]["images"] = len(many)
        for workers, context in ((4, "forkserver"), (8, "forkserver"), (12, "forkserver"), (8, "spawn"), (8, None)):
            predict_feature_files(model, vocab, many[:4 * B], batch_size=B, slots=2, workers=workers, loader_context=context)   # warm
            t0 = time.perf_counter()
            got = predict_feature_files(model, vocab, many, batch_size=B, slots=2, workers=workers, loader_context=context)
            dt = time.perf_counter() - t0
            if want is None:
                want = predict_feature_files(model, vocab, many[:8 * B], batch_size=B, slots=2)       # the one-thread loop
            assert got[:8 * B] == want and len(got) == len(many)
            name = "workers_%d_%s" % (workers, context or "fork")
            out["files_B256_loader"][name] = round(len(many) / dt, 1)
            print("[probe] files B=256, %d loader workers (%s): %.1f captions/s (one thread: %.1f)"
                  % (workers, context or "fork", len(many) / dt, base["pipelined_captions_per_s"]), file=sys.stderr, flush=True)
        for p_ in paths:
            os.remove(p_)
        os.rmdir(tmp)
    print(json.dumps(out))


if __name__ == "__main__":      # reader processes re-import this module: they must not run the probe
    main()
