"""Shared construction of parity cases: the same configuration, weights and inputs that
``tests/golden/make_goldens.py`` fed to the reference."""
import os

import numpy as np
import torch

from openviic_amd.builders import build_model
from openviic_amd.config import model_config
from openviic_amd.instance import InstanceList
from openviic_amd.utils.synthetic import (SyntheticVocab, synthetic_boxes, synthetic_features,
                                          synthetic_state_dict)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TINY = dict(d_feature=32, d_model=64, heads=4, d_kv=16, d_ff=128, layers=2, memory=5)
TINY_SHAPE = dict(B=3, N=7, V=53, T=6, k=3)
VARIANTS = ["standard_transformer", "attention_on_attention", "meshed_memory_transformer",
            "object_relation_transformer"]
FULL = dict(V=10201, T=20, N=50, D=2048)


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def tiny_case(variant, trig=False, seed=11, feature_seed=3, B=None, T=None):
    """(config, vocab, state_dict, features, boxes) of the G1/G3 tiny fixtures."""
    s = dict(TINY_SHAPE)
    if B:
        s["B"] = B
    if T:
        s["T"] = T
    vocab = SyntheticVocab(s["V"], s["T"])
    cfg = model_config(variant, trignometric_embedding=trig, device="cpu", **TINY)
    template = build_model(cfg, vocab).state_dict()
    sd = synthetic_state_dict(template, seed=seed, mode="generic", memory_dims=(TINY["d_kv"], TINY["memory"]))
    feats = synthetic_features(s["B"], s["N"], TINY["d_feature"], seed=feature_seed, ragged=True)
    boxes = synthetic_boxes(s["B"], s["N"], seed=feature_seed) if variant == "object_relation_transformer" else None
    return cfg, vocab, sd, feats, boxes


def full_case(variant, B, ragged=False):
    vocab = SyntheticVocab(FULL["V"], FULL["T"])
    cfg = model_config(variant, d_feature=FULL["D"], device="cpu")
    template = build_model(cfg, vocab).state_dict()
    sd = synthetic_state_dict(template, seed=1234, mode="reference_init")
    feats = synthetic_features(B, FULL["N"], FULL["D"], seed=0, ragged=ragged)
    boxes = synthetic_boxes(B, FULL["N"], seed=0) if variant == "object_relation_transformer" else None
    return cfg, vocab, sd, feats, boxes


DLCT_TINY = dict(B=3, n_regions=7, grid=3, d_region=32, d_grid=24, d_model=64, heads=4, d_kv=16, d_ff=128, layers=2)
DLCT_FULL = dict(B=4, n_regions=50, grid=7, d_region=2048, d_grid=1024, d_model=512, heads=8, d_kv=64, d_ff=2048, layers=3)


def dlct_case(trig, shape=None, input_seed=61):
    """(embedding cfg, encoder cfg, embedding weights, encoder weights, inputs) of the G8 fixtures (``shape=None``)
    or of a larger case with the same generators."""
    from openviic_amd.builders import build_encoder, build_vision_embedding
    from openviic_amd.config import dual_collaborative_config
    from openviic_amd.utils.synthetic import synthetic_dual_inputs
    t = shape or DLCT_TINY
    emb_cfg, enc_cfg = dual_collaborative_config(d_region=t["d_region"], d_grid=t["d_grid"], d_model=t["d_model"],
                                                 heads=t["heads"], d_kv=t["d_kv"], d_ff=t["d_ff"], layers=t["layers"],
                                                 trignometric_embedding=trig)
    emb_sd = synthetic_state_dict(build_vision_embedding(emb_cfg).state_dict(), seed=51, mode="generic")
    enc_sd = synthetic_state_dict(build_encoder(enc_cfg).state_dict(), seed=52, mode="generic")
    inputs = synthetic_dual_inputs(t["B"], t["n_regions"], t["grid"], t["d_region"], t["d_grid"], seed=input_seed)
    return emb_cfg, enc_cfg, emb_sd, enc_sd, inputs


def teacher_tokens(B, T, V, seed, with_pad=True):
    g = torch.Generator().manual_seed(seed + 77)
    tok = torch.randint(4, V, (B, T), generator=g)
    tok[:, 0] = 1
    if with_pad:
        tok[0, T - 2:] = 0
        if B > 1:
            tok[1, 2] = 0
    return tok


def device_model(cfg, vocab, sd, device="cuda"):
    """The product model on the HIP device with the case's weights."""
    cfg = cfg.clone()
    cfg.DEVICE = device
    model = build_model(cfg, vocab).eval()
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys
    return model


def batch(feats, boxes=None, tokens=None, device="cuda", field="region_features"):
    items = InstanceList()
    items[field] = feats.to(device)
    if boxes is not None:
        items["region_boxes"] = boxes.to(device)
    if tokens is not None:
        items["caption_tokens"] = tokens.to(device)
    return items


def decided_images(gaps, inner_gaps, tol):
    """Images whose every beam-boundary decision (k-th vs (k+1)-th candidate, each step) and whose
    final best-vs-second ordering had a margin above ``tol`` in the reference run.

    Margins *inside* the selected set only permute beam slots and do not change the result unless
    a later decision is itself a tie, so they are not part of the criterion.  Exact ties (margin
    0) are unspecified in the reference: ``torch.sort`` is not stable there.
    """
    margin = np.asarray(gaps).min(axis=0)                                  # (B,) over steps
    inner = np.asarray(inner_gaps)
    if inner.size:
        margin = np.minimum(margin, inner[-1, :, 0])                       # final ordering, out_size = 1
    return margin > tol


def assert_ids_match_where_decided(ids, ref_ids, gaps, inner_gaps, tol, what=""):
    """Token ids must be identical for every decided image; returns the boolean decided mask."""
    ids, ref_ids = np.asarray(ids), np.asarray(ref_ids)
    decided = decided_images(gaps, inner_gaps, tol)
    assert decided.any(), "no image has all margins above {}".format(tol)
    bad = [b for b in np.nonzero(decided)[0] if not np.array_equal(ids[b], ref_ids[b])]
    assert not bad, "{}: ids differ for decided images {}".format(what, bad)
    return decided
