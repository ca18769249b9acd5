"""SURVEY.md section 8f "next" rows that need no GPU: ids -> captions, and reference checkpoints."""
import json
import os

import numpy as np
import torch

from helpers import GOLDEN, TINY, TINY_SHAPE, tiny_case
from openviic_amd.builders import build_model
from openviic_amd.checkpoint import load_reference_checkpoint
from openviic_amd.config import model_config
from openviic_amd.utils.synthetic import SyntheticVocab, synthetic_features
from openviic_amd.vocab import WordVocab, captions_from_ids, collapse_repeated_words
from oracle.captioner import OracleCaptioner


def test_decode_caption_matches_reference_strings():
    g = json.load(open(os.path.join(GOLDEN, "g6_decode_caption.json")))
    vocab = WordVocab(g["itos"], max_caption_length=8)
    assert (vocab.padding_idx, vocab.bos_idx, vocab.eos_idx, vocab.unk_idx, len(vocab)) == (0, 1, 2, 3, 53)
    ids = torch.tensor(g["ids"])
    assert vocab.decode_caption(ids, join_words=True) == g["joined"]
    assert vocab.decode_caption(ids, join_words=False) == g["split"]
    assert captions_from_ids(vocab, ids) == g["collapsed"]
    assert collapse_repeated_words(["a", "a", "b", "a"]) == "a b a"
    enc = vocab.encode_caption(["w00", "nope"])
    assert enc.tolist() == [1, 4, 3, 2, 0, 0, 0, 0]


def test_reference_checkpoint_loads_key_for_key_and_oracle_reproduces_its_output():
    for variant in ("standard_transformer", "meshed_memory_transformer"):
        path = os.path.join(GOLDEN, "g7_reference_checkpoint_%s.pth" % variant)
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        vocab = SyntheticVocab(TINY_SHAPE["V"], TINY_SHAPE["T"])
        cfg = model_config(variant, device="cpu", **TINY)
        model = build_model(cfg, vocab)
        result = load_reference_checkpoint(model, path, strict=True)          # every key, every shape
        assert not result.missing_keys and not result.unexpected_keys
        for k, v in ckpt["state_dict"].items():
            assert torch.equal(model.state_dict()[k], v), k
        feats = synthetic_features(TINY_SHAPE["B"], TINY_SHAPE["N"], TINY["d_feature"], seed=3, ragged=True)
        ids, logp = OracleCaptioner(cfg, ckpt["state_dict"], len(vocab), vocab.max_caption_length).beam_search(feats, TINY_SHAPE["k"])
        assert torch.equal(ids, ckpt["beam_ids"])
        np.testing.assert_allclose(logp.numpy(), ckpt["beam_logp"].numpy(), rtol=1e-5, atol=5e-6)
