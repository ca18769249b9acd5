import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import *
for variant in ["standard_transformer", "attention_on_attention", "meshed_memory_transformer"]:
    g = golden("g2_full_%s.npz" % variant)
    cfg, vocab, sd, feats16, _ = full_case(variant, 16)
    model = device_model(cfg, vocab, sd)
    for B, k in [(4, 1), (4, 5), (16, 5)]:
        ids, logp = model.beam_search(batch(feats16[:B]), batch_size=B, beam_size=k)
        p = "B%d_k%d_" % (B, k)
        same = (ids.cpu().numpy() == g[p + "ids"]).all(axis=1)
        gap = g[p + "gap"].min(axis=0)
        fin = g[p + "inner_gap"][-1, :, 0] if k > 1 else np.full(B, np.inf)
        inner = g[p + "inner_gap"].min(axis=(0, 2)) if k > 1 else np.full(B, np.inf)
        print(variant, p, "same", same.astype(int), "\n  boundary gap", gap, "\n  final gap", fin, "\n  inner", inner)
        d = np.abs(logp.cpu().numpy() - g[p + "logp"])[same]
        print("  max logp err on same:", d.max() if d.size else None)
g = golden("g3_forced_eos_pad.npz")
cfg, vocab, sd, feats, _ = tiny_case("standard_transformer", seed=21, feature_seed=8, B=6, T=8)
sd["decoder.fc.weight"] = torch.from_numpy(g["decoder.fc.weight"])
model = device_model(cfg, vocab, sd)
ids, logp, allp = model.beam_search(batch(feats), batch_size=6, beam_size=3, out_size=3, return_probs=True)
print("g3 ids\n", ids.cpu().numpy(), "\nref\n", g["ids"])
print("gap", g["gap"].T, "\ninner", g["inner_gap"].transpose(1,0,2))
print("score ref", g["score"].transpose(1,0,2))
