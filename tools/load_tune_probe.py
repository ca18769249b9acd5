"""How much would ranking GEMM tilings under the REAL load buy?  (speed only: every tiling of a K-order class gives the same bits)

The engine ranks tilings per shape with gridDim.z co-running copies of the same product (objective c).  This probe instead
walks the decode / encoder shapes in order of their time share and, for each, tries every tiling of the shape's class while the
whole caption batch runs on four streams -- coordinate descent on the measured captions/s.

    python tools/load_tune_probe.py [--streams 4] [--steps 24] [--rounds 1]
"""
import argparse
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=4)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--rounds", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256)
    args = ap.parse_args()
    from bench import D_FEAT, N_REGIONS, T, V                                    # the bench's workload constants
    from openviic_amd import native
    from openviic_amd.builders import build_model
    from openviic_amd.config import model_config
    from openviic_amd.engine import CaptionEngine
    from openviic_amd.utils.synthetic import SyntheticVocab, synthetic_features, synthetic_state_dict
    lib = native.load()
    lib.ovc_profile_kernel_name.restype = ctypes.c_char_p
    device = torch.device("cuda", 0)
    vocab = SyntheticVocab(V, T)
    model = build_model(model_config("standard_transformer", d_feature=D_FEAT, device=str(device)), vocab).eval()
    model.load_state_dict(synthetic_state_dict(model.state_dict(), seed=1234, mode="reference_init"), strict=False)
    B, k = args.batch, 5
    feats = synthetic_features(B, N_REGIONS, D_FEAT, seed=0).to(device)
    objective = min(args.streams, 4) if args.streams >= 3 else 1
    engine = CaptionEngine(model, tune_concurrency=objective)
    streams = [torch.cuda.Stream(device=device) for _ in range(args.streams)]

    def rate(steps):
        with torch.no_grad():
            for i in range(3 * len(streams)):                 # plain pass, capture, first replay on every stream
                with torch.cuda.stream(streams[i % len(streams)]):
                    engine.beam_search(feats, None, B, k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                with torch.cuda.stream(streams[i % len(streams)]):
                    engine.beam_search(feats, None, B, k)
            torch.cuda.synchronize()
        return B * steps / (time.perf_counter() - t0)

    base = rate(args.steps)
    base2 = rate(args.steps)
    print("baseline (objective %d table): %.0f / %.0f captions/s" % (objective, base, base2), flush=True)
    tilings = []
    t = 0
    while lib.ovc_profile_kernel_name(t):
        tilings.append((t, lib.ovc_profile_kernel_name(t).decode(), None))
        t += 1
    shapes = engine.gemm_shapes(B, N_REGIONS, k)
    # weight of a shape ~ its FLOPs x how often it runs (decode shapes: 19 steps x 3 layers)
    def weight(sh):
        M, n, nseg, K = sh[:4]
        reps = 57 if M in (B * k,) else (3 if M == B else 1)
        return 2.0 * M * n * nseg * K * reps
    shapes = sorted(shapes, key=weight, reverse=True)
    best_rate = max(base, base2)
    for rnd in range(args.rounds):
        for sh in shapes[:12]:
            current = lib.ovc_gemm_tuned_get(*sh[:6], objective, 0)
            results = []
            for t, name, _ in tilings:
                if lib.ovc_gemm_tuned_set(*sh[:6], objective, t) != 0:          # another class, or does not fit the shape
                    continue
                engine.release()
                results.append((rate(args.steps), t, name))
            results.sort(reverse=True)
            pick = results[0]
            # adopt only a clear win over the current choice's own measurement in this sweep
            cur = [r for r in results if r[1] == current]
            keep = current
            if cur and pick[0] > cur[0][0] * 1.004:
                keep = pick[1]
            lib.ovc_gemm_tuned_set(*sh[:6], objective, keep)
            engine.release()
            print("shape %s: current %s -> keep %s | %s" % (sh, current, keep, ", ".join("%d:%.0f" % (r[1], r[0]) for r in results)), flush=True)
        final = [rate(args.steps) for _ in range(3)]
        print("round %d: %s captions/s (baseline %.0f / %.0f)" % (rnd, " / ".join("%.0f" % f for f in final), base, base2), flush=True)


if __name__ == "__main__":
    main()
