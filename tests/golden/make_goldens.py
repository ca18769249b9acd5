#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the reference implementation itself.

Run in the build container only (the reference checkout does not exist on the GPU box):

    python tests/golden/make_goldens.py [--reference /root/reference]

The reference is imported, never copied: its model classes are built through its own
``build_model`` from the ``MODEL`` node produced by ``openviic_amd.config.model_config`` (same keys
as the reference yaml files), loaded with the deterministic weights of
``openviic_amd.utils.synthetic`` and run on the deterministic synthetic inputs.  Only inputs that
cannot be regenerated and the reference's outputs are stored (``*.npz``, float32/int64 arrays).

Two inert stand-ins are placed in ``sys.modules`` for import-time dependencies that are absent in
this image and unused on the hot path: ``termcolor`` (logger colours) and ``cv2`` (image loading).

Fixture families:
  G1  tiny configuration, every variant: per-module intermediates, teacher-forced log-probs,
      beam search with ``return_probs`` and ``out_size=k``
  G2  full-size configurations (d=512, N=50, d_feat=2048, V=10201, T=20): ids, log-probs and
      per-decision selection gaps for greedy / beam-5 decoding
  G3  tiny configuration with a sharpened vocabulary projection that forces <eos> and <pad> (standard and
      meshed-memory decoders)
  G4  operator-level geometry cross-attention with nq != nk and a per-query mask (DLCT form)
  G5  box relation embedding for both ``trignometric_embedding`` values
  G6  ids -> caption strings through the reference's ``Vocab.decode_caption`` + duplicate collapse
  G7  a checkpoint written by the reference (its own initialisation) with its beam-search output
  G8  dual-collaborative (DLCT) embedding + encoder: the reference's own sub-modules composed
      harness-side with the three mask-shape repairs listed in ``g8_dlct_encoder``
  G9  the prediction loop's input and output sides: ragged per-image feature dicts collated by the reference's
      ``InstanceList`` (utils/instance.py:33-55,156-171 via data_utils/utils.py:120-121), and the caption strings
      the reference produces for them end to end (G7 checkpoint -> beam_search(out_size=1) -> Vocab.decode_caption
      -> groupby collapse, trainers/vi_trainer.py:242-252)
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from openviic_amd.config import model_config                      # noqa: E402
from openviic_amd.utils.synthetic import (SyntheticVocab, synthetic_boxes,   # noqa: E402
                                          synthetic_features, synthetic_state_dict)

TINY = dict(d_feature=32, d_model=64, heads=4, d_kv=16, d_ff=128, layers=2, memory=5)
TINY_SHAPE = dict(B=3, N=7, V=53, T=6, k=3)
VARIANTS = ["standard_transformer", "attention_on_attention", "meshed_memory_transformer",
            "object_relation_transformer"]


def import_reference(path):
    sys.dont_write_bytecode = True
    sys.path.insert(0, path)
    termcolor = types.ModuleType("termcolor")
    termcolor.colored = lambda s, *a, **k: s
    sys.modules.setdefault("termcolor", termcolor)
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    import models  # noqa: F401  (populates the reference registries)
    from builders.model_builder import build_model
    from utils.instance import InstanceList
    from models.modules.beam_search import BeamSearch
    from models.modules.attentions import MultiHeadAttention
    from models.utils import box_relational_embedding
    return dict(build_model=build_model, InstanceList=InstanceList, BeamSearch=BeamSearch,
                MultiHeadAttention=MultiHeadAttention, box_relational_embedding=box_relational_embedding)


def build_reference(ref, cfg, vocab, seed, mode, memory_dims=None):
    cfg = cfg.clone()
    cfg.DEVICE = "cpu"
    model = ref["build_model"](cfg, vocab).eval()
    weights = synthetic_state_dict(model.state_dict(), seed=seed, mode=mode, memory_dims=memory_dims)
    missing = model.load_state_dict(weights, strict=False)
    assert not missing.unexpected_keys, missing.unexpected_keys
    if cfg.ARCHITECTURE == "ObjectRelationTransformer":
        # The shipped class passes one Instance to GeometricEncoder.forward, which takes
        # (features, boxes, padding_mask) -- a TypeError.  Wire the
        # reference's own sub-modules by keyword, harness-side.
        def encoder_forward(input_features, _m=model):
            feats, mask = _m.vision_embedding(input_features.region_features)
            return _m.encoder(features=feats, boxes=input_features.region_boxes, padding_mask=mask), mask
        model.encoder_forward = encoder_forward
    return model


def make_inputs(ref, B, N, d_feature, seed, ragged, boxes):
    items = ref["InstanceList"]()
    items.region_features = synthetic_features(B, N, d_feature, seed=seed, ragged=ragged)
    if boxes:
        items.region_boxes = synthetic_boxes(B, N, seed=seed)
    return items


class SelectRecorder:
    """Wraps the reference's ``BeamSearch.select`` to record decision margins."""

    def __init__(self, ref):
        self.cls = ref["BeamSearch"]
        self.orig = self.cls.select
        self.gap, self.inner, self.chosen, self.score = [], [], [], []

    def __enter__(self):
        rec = self

        def select(this, candidate_logprob):
            idx, val = rec.orig(this, candidate_logprob)
            flat = candidate_logprob.view(this.b_s, -1)
            k = this.beam_size
            top = torch.topk(flat, min(k + 1, flat.shape[1]), dim=-1).values
            rec.gap.append((top[:, k - 1] - top[:, k]).clone() if top.shape[1] > k
                           else torch.full((this.b_s,), float("inf")))
            rec.inner.append((top[:, :k - 1] - top[:, 1:k]).clone())
            rec.chosen.append(idx.clone())
            rec.score.append(val.clone())
            return idx, val
        self.cls.select = select
        return self

    def __exit__(self, *exc):
        self.cls.select = self.orig

    def arrays(self, prefix):
        return {prefix + "gap": torch.stack(self.gap).numpy(),
                prefix + "inner_gap": torch.stack(self.inner).numpy(),
                prefix + "chosen": torch.stack(self.chosen).numpy(),
                prefix + "score": torch.stack(self.score).numpy()}


def teacher_tokens(B, T, V, seed, with_pad):
    g = torch.Generator().manual_seed(seed + 77)
    tok = torch.randint(4, V, (B, T), generator=g)
    tok[:, 0] = 1
    if with_pad:
        tok[0, T - 2:] = 0              # trailing padding
        if B > 1:
            tok[1, 2] = 0               # <pad> in the middle of a sequence
    return tok


def hook_intermediates(model, store):
    handles = []

    def keep(name):
        def fn(_module, _inp, out):
            store[name] = (out[0] if isinstance(out, tuple) else out).detach().clone().numpy()
        return fn
    handles.append(model.vision_embedding.register_forward_hook(keep("feature_proj")))
    for i, layer in enumerate(model.encoder.layers):
        handles.append(layer.mhatt.register_forward_hook(keep("enc%d_mhatt" % i)))
        handles.append(layer.register_forward_hook(keep("enc%d_out" % i)))
    for i, layer in enumerate(model.decoder.layers):
        handles.append(layer.self_attn.register_forward_hook(keep("dec%d_self" % i)))
        handles.append(layer.register_forward_hook(keep("dec%d_out" % i)))
    return handles


def g1_tiny(ref, out_dir, variant, trig=False, tag=None):
    s = TINY_SHAPE
    vocab = SyntheticVocab(s["V"], s["T"])
    cfg = model_config(variant, trignometric_embedding=trig, **TINY)
    model = build_reference(ref, cfg, vocab, seed=11, mode="generic", memory_dims=(TINY["d_kv"], TINY["memory"]))
    boxes = variant == "object_relation_transformer"
    items = make_inputs(ref, s["B"], s["N"], TINY["d_feature"], seed=3, ragged=True, boxes=boxes)
    data = {}
    with torch.no_grad():
        enc, mask = model.encoder_forward(items)
        data["enc_out"], data["enc_mask"] = enc.numpy(), mask.numpy()
        items.caption_tokens = teacher_tokens(s["B"], s["T"], s["V"], seed=5, with_pad=True)
        data["caption_tokens"] = items.caption_tokens.numpy()
        handles = hook_intermediates(model, data)
        data["forward_logp"] = model(items).numpy()
        for hnd in handles:
            hnd.remove()
        for k in (1, s["k"]):
            with SelectRecorder(ref) as rec:
                ids, logp, allp = model.beam_search(items, batch_size=s["B"], beam_size=k, out_size=k,
                                                    return_probs=True)
            data["beam%d_ids" % k], data["beam%d_logp" % k] = ids.numpy(), logp.numpy()
            data["beam%d_all" % k] = allp.numpy()
            data.update(rec.arrays("beam%d_" % k))
        ids1, logp1 = model.beam_search(items, batch_size=s["B"], beam_size=s["k"], out_size=1)
        data["beam_out1_ids"], data["beam_out1_logp"] = ids1.numpy(), logp1.numpy()
    name = "g1_tiny_%s.npz" % (tag or variant)
    np.savez_compressed(os.path.join(out_dir, name), **data)
    print("wrote", name, {k: v.shape for k, v in data.items() if k.startswith("beam")})


def g2_full(ref, out_dir, variant, batches=(4, 16, 48)):
    V, T, N, D = 10201, 20, 50, 2048
    vocab = SyntheticVocab(V, T)
    cfg = model_config(variant, d_feature=D)
    model = build_reference(ref, cfg, vocab, seed=1234, mode="reference_init")
    boxes = variant == "object_relation_transformer"
    data = {}
    with torch.no_grad():
        for B, k in [(batches[0], 1), (batches[0], 5), (batches[1], 5), (batches[2], 5)]:
            items = make_inputs(ref, B, N, D, seed=0, ragged=False, boxes=boxes)
            with SelectRecorder(ref) as rec:
                ids, logp = model.beam_search(items, batch_size=B, beam_size=k, out_size=1)
            p = "B%d_k%d_" % (B, k)
            data[p + "ids"], data[p + "logp"] = ids.numpy(), logp.numpy()
            data.update(rec.arrays(p))
        B = batches[0]
        items = make_inputs(ref, B, N, D, seed=0, ragged=True, boxes=boxes)
        items.caption_tokens = teacher_tokens(B, T, V, seed=9, with_pad=True)
        full = model(items)
        data["fwd_tokens"] = items.caption_tokens.numpy()
        data["fwd_max"], data["fwd_argmax"] = [x.numpy() for x in full.max(-1)]
        data["fwd_sample"] = full[:, :, ::97].contiguous().numpy()
        enc, _ = model.encoder_forward(items)
        data["enc_sample"] = enc.reshape(B, -1, enc.shape[-1])[:, ::7, ::5].contiguous().numpy()
    name = "g2_full_%s.npz" % variant
    np.savez_compressed(os.path.join(out_dir, name), **data)
    print("wrote", name)


def g3_forced(ref, out_dir, variant="standard_transformer"):
    """<eos> and <pad> are (almost) never emitted under random weights; edit their rows.

    The edited ``decoder.fc.weight`` is stored in the fixture."""
    s = dict(TINY_SHAPE, B=6, T=8)
    vocab = SyntheticVocab(s["V"], s["T"])
    cfg = model_config(variant, **TINY)
    model = build_reference(ref, cfg, vocab, seed=21, mode="generic", memory_dims=(TINY["d_kv"], TINY["memory"]))
    with torch.no_grad():
        w = model.decoder.fc.weight
        w[vocab.eos_idx] = 1.05 * w[8]       # <eos> tracks a frequently chosen word
        w[vocab.padding_idx] *= 2.0          # <pad> gets emitted by live beams
    items = make_inputs(ref, s["B"], s["N"], TINY["d_feature"], seed=8, ragged=True, boxes=False)
    data = {"decoder.fc.weight": model.decoder.fc.weight.detach().clone().numpy()}
    with torch.no_grad():
        with SelectRecorder(ref) as rec:
            ids, logp, allp = model.beam_search(items, batch_size=s["B"], beam_size=s["k"], out_size=s["k"],
                                                return_probs=True)
    data["ids"], data["logp"], data["all"] = ids.numpy(), logp.numpy(), allp.numpy()
    data.update(rec.arrays(""))
    n_eos, n_pad = int((ids == 2).sum()), int((ids == 0).sum())
    assert n_eos > 0 and n_pad > 0, (n_eos, n_pad)
    name = "g3_forced_eos_pad.npz" if variant == "standard_transformer" else "g3_forced_eos_pad_%s.npz" % variant
    np.savez_compressed(os.path.join(out_dir, name), **data)
    print("wrote %s  eos=%d pad=%d of %d" % (name, n_eos, n_pad, ids.numel()))


def g4_dlct_operator(ref, out_dir):
    from openviic_amd.config import ConfigNode
    B, nq, nk, d, h = 2, 50, 99, 512, 8
    att = ConfigNode(dict(ARCHITECTURE="AugmentedGeometryScaledDotProductAttention", HEAD=h, D_MODEL=d,
                          D_KEY=64, D_VALUE=64, D_FF=2048, USE_AOA=False, CAN_BE_STATEFUL=False, DROPOUT=0.1))
    mha = ref["MultiHeadAttention"](att).eval()
    sd = synthetic_state_dict({"x." + k: v for k, v in mha.state_dict().items()}, seed=31, mode="generic")
    mha.load_state_dict({k[2:]: v for k, v in sd.items()})
    g = torch.Generator().manual_seed(41)
    q = torch.randn(B, nq, d, generator=g)
    kv = torch.randn(B, nk, d, generator=g)
    geo = torch.rand(B, h, nq, nk, generator=g) * 2.0 - 0.5          # negatives exercise clamp(1e-6)
    mask = torch.rand(B, 1, nq, nk, generator=g) < 0.3
    mask[:, :, :, 0] = False                                          # no fully masked row
    with torch.no_grad():
        out = mha(queries=q, keys=kv, values=kv, padding_mask=None, attention_mask=mask,
                  relative_geometry_weights=geo)
    np.savez_compressed(os.path.join(out_dir, "g4_dlct_cross_attention.npz"),
                        queries=q.numpy(), keys=kv.numpy(), geometry=geo.numpy(), mask=mask.numpy(),
                        out=out.numpy())
    print("wrote g4_dlct_cross_attention.npz")


def g5_box_relation(ref, out_dir):
    boxes = synthetic_boxes(2, 9, seed=5)
    boxes[1, 3] = boxes[1, 2]                                         # identical boxes -> clamp(1e-3) branch
    data = {"boxes": boxes.numpy()}
    for trig in (False, True):
        emb = ref["box_relational_embedding"](boxes, dim_g=16 if trig else 4, trignometric_embedding=trig)
        data["trig" if trig else "plain"] = emb.numpy()
    np.savez_compressed(os.path.join(out_dir, "g5_box_relation.npz"), **data)
    print("wrote g5_box_relation.npz")


def g6_decode_caption(ref, out_dir):
    """ids -> words (stop at <eos>, drop specials) and the trainer's duplicate-word collapse
    (data_utils/vocab.py:104-122, trainers/vi_trainer.py:247-252)."""
    import itertools
    import json
    from data_utils.vocab import Vocab
    vocab = object.__new__(Vocab)          # the constructor needs the dataset json; only these fields are read
    vocab.itos = ["<pad>", "<bos>", "<eos>", "<unk>"] + ["w%02d" % i for i in range(49)]
    vocab.specials = vocab.itos[:4]
    vocab.eos_idx = 2
    forced = np.load(os.path.join(out_dir, "g3_forced_eos_pad.npz"))["ids"].reshape(-1, 8)
    extra = np.array([[4, 4, 4, 5, 5, 2, 9, 9], [1, 3, 7, 7, 0, 7, 2, 2], [2, 0, 0, 0, 0, 0, 0, 0], [6, 7, 8, 9, 10, 11, 12, 13]])
    ids = torch.from_numpy(np.concatenate([forced, extra]))
    joined = vocab.decode_caption(ids, join_words=True)
    split = vocab.decode_caption(ids, join_words=False)
    collapsed = [" ".join(k for k, _ in itertools.groupby(words)) for words in split]
    with open(os.path.join(out_dir, "g6_decode_caption.json"), "w") as f:
        json.dump({"itos": vocab.itos, "ids": ids.tolist(), "joined": joined, "split": split, "collapsed": collapsed}, f)
    print("wrote g6_decode_caption.json (%d captions)" % len(joined))


def g7_reference_checkpoint(ref, out_dir):
    """A state_dict exactly as the reference's own constructors and torch.save produce it
    (trainers/base_trainer.py:138-153 stores it under 'state_dict')."""
    s = TINY_SHAPE
    for variant in ("standard_transformer", "meshed_memory_transformer"):
        cfg = model_config(variant, **TINY)
        cfg.DEVICE = "cpu"
        torch.manual_seed(99)
        model = ref["build_model"](cfg, SyntheticVocab(s["V"], s["T"])).eval()
        items = make_inputs(ref, s["B"], s["N"], TINY["d_feature"], seed=3, ragged=True, boxes=False)
        with torch.no_grad():
            ids, logp = model.beam_search(items, batch_size=s["B"], beam_size=s["k"], out_size=1)
        torch.save({"state_dict": model.state_dict(), "epoch": 3, "beam_ids": ids, "beam_logp": logp},
                   os.path.join(out_dir, "g7_reference_checkpoint_%s.pth" % variant))
        print("wrote g7_reference_checkpoint_%s.pth (%d tensors)" % (variant, len(model.state_dict())))


def g9_instances(seed=17):
    """The per-image feature dicts of G9, exactly as tests re-create them (numpy float32 / float64 / int64 arrays with
    ragged first dimensions, a torch tensor field, non-tensor fields)."""
    rng = np.random.default_rng(seed)
    out = []
    for image_id, n in enumerate((5, 7, 3, 6)):
        out.append(dict(
            filename="img_%d.jpg" % image_id, image_id=100 + image_id, captions=["caption %d a" % image_id, "caption %d b" % image_id],
            region_features=rng.standard_normal((n, TINY["d_feature"])).astype(np.float32),
            region_boxes=np.concatenate([rng.random((n, 2)) * 0.5, 0.5 + rng.random((n, 2)) * 0.5], 1).astype(np.float32),
            grid_features=torch.from_numpy(rng.standard_normal((4, 6)).astype(np.float32)),      # torch field, not ragged
            region_scores=rng.random((n, 1)),                                                    # float64, ragged
            region_labels=rng.integers(0, 9, (n, 2)),                                            # int64, ragged: padding promotes it
        ))
    return out


def g9_prediction_loop(ref, out_dir):
    """Input side: ``collate_fn(samples) = InstanceList(samples)`` on ragged Instances.  Output side: the reference's
    prediction loop on the collated batch with the weights of the G7 checkpoint (its own initialisation)."""
    import itertools
    import json
    from data_utils.vocab import Vocab
    from utils.instance import Instance
    s = TINY_SHAPE
    samples = g9_instances()
    batch = ref["InstanceList"]([Instance(**d) for d in samples])
    arrays = {}
    for key in ("region_features", "region_boxes", "grid_features", "region_scores", "region_labels"):
        arrays[key] = batch[key].numpy()
        arrays[key + "_dtype"] = np.array(str(batch[key].dtype))
    assert batch.batch_size == 4
    lists = {k: batch[k] for k in ("filename", "image_id", "captions")}
    # the reference's own prediction loop on that batch (trainers/vi_trainer.py:242-252)
    variant = "standard_transformer"
    ckpt = torch.load(os.path.join(out_dir, "g7_reference_checkpoint_%s.pth" % variant), map_location="cpu", weights_only=True)
    cfg = model_config(variant, **TINY)
    cfg.DEVICE = "cpu"
    model = ref["build_model"](cfg, SyntheticVocab(s["V"], s["T"])).eval()
    model.load_state_dict(ckpt["state_dict"], strict=True)
    vocab = object.__new__(Vocab)
    vocab.itos = ["<pad>", "<bos>", "<eos>", "<unk>"] + ["w%02d" % i for i in range(s["V"] - 4)]
    vocab.specials = vocab.itos[:4]
    vocab.eos_idx, vocab.max_caption_length = 2, s["T"]
    recorded = {}
    import models.modules.beam_search as bs_mod                     # record the decision margins of this run
    orig_select = bs_mod.BeamSearch.select
    def select(self, candidate_logprob):
        flat = candidate_logprob.view(self.b_s, -1)
        top = torch.sort(flat, -1, descending=True)[0]
        recorded.setdefault("gap", []).append((top[:, self.beam_size - 1] - top[:, min(self.beam_size, top.shape[1] - 1)]).numpy().copy())
        return orig_select(self, candidate_logprob)
    bs_mod.BeamSearch.select = select
    try:
        with torch.no_grad():
            outs, _ = model.beam_search(batch, batch_size=batch.batch_size, beam_size=s["k"], out_size=1)
    finally:
        bs_mod.BeamSearch.select = orig_select
    caps_gen = vocab.decode_caption(outs.contiguous().view(-1, vocab.max_caption_length), join_words=False)
    gens = [" ".join(k for k, g in itertools.groupby(gen_i)) for gen_i in caps_gen]
    np.savez_compressed(os.path.join(out_dir, "g9_collated_batch.npz"), beam_ids=outs.numpy(),
                        gap=np.stack(recorded["gap"]), **arrays)
    with open(os.path.join(out_dir, "g9_prediction_loop.json"), "w") as f:
        json.dump({"itos": vocab.itos, "lists": lists, "gens": gens, "beam_size": s["k"]}, f)
    print("wrote g9_collated_batch.npz / g9_prediction_loop.json:", gens)


DLCT_TINY = dict(B=3, n_regions=7, grid=3, d_region=32, d_grid=24, d_model=64, heads=4, d_kv=16, d_ff=128, layers=2)


def dlct_config(t, trig):
    from openviic_amd.config import dual_collaborative_config
    return dual_collaborative_config(d_region=t["d_region"], d_grid=t["d_grid"], d_model=t["d_model"], heads=t["heads"],
                                     d_kv=t["d_kv"], d_ff=t["d_ff"], layers=t["layers"], trignometric_embedding=trig)


def dlct_inputs(t, seed):
    from openviic_amd.utils.synthetic import synthetic_dual_inputs
    return synthetic_dual_inputs(t["B"], t["n_regions"], t["grid"], t["d_region"], t["d_grid"], seed=seed)


def g8_dlct_encoder(ref, out_dir):
    """The reference ships ``GeometricDualFeatureEmbedding`` and ``DualCollaborativeLevelEncoder``
    (vision_embeddings.py:46-71, encoders.py:115-211) but neither runs.  The
    fixture is produced by the reference's own sub-modules -- projections, ``get_combine_masks``,
    ``box_relational_embedding``, ``fc_gs``, layer norms, positional embedding, the self-attention
    ``EncoderLayer``s as they are, and ``mhatt`` / ``pwff`` of the cross layers -- composed here in the order of
    the reference's ``forward`` with three repairs:
      1. ``get_combine_masks`` returns (B,1,1,n,g*g); the embedding treats it as 4-D: use (B,1,n,g*g);
      2. the key-padding masks (B,1,1,n) are expanded over the query dimension before they are
         concatenated with the region<->grid visibility masks;
      3. a cross ``EncoderLayer`` clears padded rows with ``padding_mask.squeeze(1).squeeze(1)``, which cannot
         broadcast for a per-query mask: clear the rows of the *query* side's padding mask instead.
    """
    from models.modules.encoders import DualCollaborativeLevelEncoder
    from models.modules.vision_embeddings import GeometricDualFeatureEmbedding
    from models.utils import box_relational_embedding, generate_padding_mask, get_combine_masks
    import torch.nn.functional as F
    t = DLCT_TINY
    for trig in (False, True):
        emb_cfg, enc_cfg = dlct_config(t, trig)
        emb = GeometricDualFeatureEmbedding(emb_cfg).eval()
        enc = DualCollaborativeLevelEncoder(enc_cfg).eval()
        emb.load_state_dict(synthetic_state_dict(emb.state_dict(), seed=51, mode="generic"))
        enc.load_state_dict(synthetic_state_dict(enc.state_dict(), seed=52, mode="generic"))
        region, region_boxes, grid, grid_boxes = dlct_inputs(t, seed=61)
        B, n, gg = t["B"], t["n_regions"], t["grid"] ** 2
        store = {}
        with torch.no_grad():
            # ---- embedding (vision_embeddings.py:56-71) ----
            region_mask = generate_padding_mask(region, padding_idx=0)
            grid_mask = generate_padding_mask(grid, padding_idx=0)
            r2g = get_combine_masks(region_boxes, t["grid"]).reshape(B, 1, n, gg)                      # repair 1
            region2all = torch.cat([region_mask.expand(B, 1, n, n), r2g], dim=-1)                      # repair 2
            grid2all = torch.cat([r2g.permute(0, 1, 3, 2), grid_mask.expand(B, 1, gg, gg)], dim=-1)
            rf, gf = emb.region_proj(region), emb.grid_proj(grid)
            store.update(region_embedded=rf, grid_embedded=gf, region_mask=region_mask, grid_mask=grid_mask,
                         region2all_mask=region2all, grid2all_mask=grid2all)
            # ---- encoder (encoders.py:153-211) ----
            boxes = torch.cat([region_boxes, grid_boxes], dim=1)
            rel = box_relational_embedding(boxes, dim_g=enc.d_g, trignometric_embedding=enc.trignometric_embedding)
            nk = rel.shape[1]
            w = torch.cat([fc(rel.view(-1, enc.d_g)).view(B, 1, nk, nk) for fc in enc.fc_gs], dim=1)
            w = F.relu(w)
            rf = enc.layer_norm_region(rf) + enc.pos_embedding(rf)
            gf = enc.layer_norm_grid(gf) + enc.pos_embedding(gf)
            for li, (l_r, l_g, l_r2g, l_g2r) in enumerate(zip(enc.layers_region, enc.layers_grid, enc.region2grid, enc.grid2region)):
                rf = l_r(queries=rf, keys=rf, values=rf, relative_geometry_weights=w[:, :, :n, :n],
                         padding_mask=region_mask, attention_mask=region_mask)
                gf = l_g(queries=gf, keys=gf, values=gf, relative_geometry_weights=w[:, :, n:, n:],
                         padding_mask=grid_mask, attention_mask=grid_mask)
                combined = torch.cat([rf, gf], dim=1)
                combined = combined + enc.pos_embedding(combined)

                def cross(layer, q, geo, mask, q_pad):
                    att = layer.mhatt(queries=q, keys=combined, values=combined, padding_mask=mask,
                                      attention_mask=mask, relative_geometry_weights=geo)
                    return layer.pwff(att).masked_fill(q_pad[:, 0, 0, :, None], 0)                     # repair 3
                rf = cross(l_r2g, rf, w[:, :, :n, :], region2all, region_mask)
                gf = cross(l_g2r, gf, w[:, :, n:, :], grid2all, grid_mask)
                store["layer%d_region" % li], store["layer%d_grid" % li] = rf, gf
            store["out"] = torch.cat([rf, gf], dim=1)
            store["padding_mask"] = torch.cat([region_mask, grid_mask], dim=-1)
            store["geometry_weights"] = w
        data = {k: v.numpy() for k, v in store.items()}
        data.update(region_features=region.numpy(), region_boxes=region_boxes.numpy(), grid_features=grid.numpy(),
                    grid_boxes=grid_boxes.numpy())
        # cell look-up at float32 values next to the (float64) cell edges, on a 10 x 10 grid
        edges = torch.tensor([0.0, 0.1, 0.3, 0.7, 0.9, 0.5, 0.2, 0.6], dtype=torch.float32)
        near = torch.stack([edges, torch.nextafter(edges, torch.tensor(-1.0)), torch.nextafter(edges, torch.tensor(2.0))]).flatten()
        gen = torch.Generator().manual_seed(71)
        pick = lambda: near[torch.randint(0, near.numel(), (2, 12), generator=gen)]      # noqa: E731
        lo_x, lo_y, hi_x, hi_y = pick(), pick(), pick(), pick()
        edge_boxes = torch.stack([lo_x, lo_y, torch.maximum(lo_x, hi_x), torch.maximum(lo_y, hi_y)], dim=-1).clamp(0, 0.95)
        edge_boxes[1, 0] = torch.tensor([0.5, 0.5, 0.2, 0.9])          # x_max < x_min: nothing visible
        edge_boxes[1, 1] = torch.tensor([-0.2, 0.3, 0.4, 0.3])         # below the first edge: cell 0
        data.update(edge_boxes=edge_boxes.numpy(), edge_mask_g10=get_combine_masks(edge_boxes, 10).reshape(2, 1, 12, 100).numpy())
        name = "g8_dlct_encoder%s.npz" % ("_trig" if trig else "")
        np.savez_compressed(os.path.join(out_dir, name), **data)
        print("wrote %s  (visible region->grid cells: %d of %d)" % (name, int((~r2g).sum()), r2g.numel()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    torch.set_num_threads(8)
    ref = import_reference(args.reference)
    want = set(args.only.split(",")) if args.only else None

    def on(tag):
        return want is None or tag in want
    if on("g1"):
        for v in VARIANTS:
            g1_tiny(ref, HERE, v)
        g1_tiny(ref, HERE, "object_relation_transformer", trig=True, tag="object_relation_transformer_trig")
    if on("g3"):
        g3_forced(ref, HERE)
        g3_forced(ref, HERE, "meshed_memory_transformer")
    if on("g4"):
        g4_dlct_operator(ref, HERE)
    if on("g5"):
        g5_box_relation(ref, HERE)
    if on("g6"):
        g6_decode_caption(ref, HERE)
    if on("g7"):
        g7_reference_checkpoint(ref, HERE)
    if on("g8"):
        g8_dlct_encoder(ref, HERE)
    if on("g9"):
        g9_prediction_loop(ref, HERE)
    if on("g2"):
        for v in VARIANTS:
            g2_full(ref, HERE, v)


if __name__ == "__main__":
    main()
