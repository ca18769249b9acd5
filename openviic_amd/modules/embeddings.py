"""Vision / text / positional embeddings, host side.

``FeatureEmbedding``: reference ``models/modules/vision_embeddings.py:8-20``;
``UsualEmbedding``: ``models/modules/text_embeddings.py:8-31`` (``WORD_EMBEDDING: null`` branch);
``SinusoidPositionalEmbedding``: ``models/modules/pos_embeddings.py:39-72``.
"""
import math

import torch
from torch import nn

from .. import ops
from ..builders.text_embedding_builder import META_TEXT_EMBEDDING
from ..builders.vision_embedding_builder import META_VISION_EMBEDDING


@META_VISION_EMBEDDING.register()
class FeatureEmbedding(nn.Module):
    """Linear projection of region/grid features + padding mask (rows whose sum is exactly 0)."""

    def __init__(self, config):
        super().__init__()
        self.proj = nn.Linear(config.D_FEATURE, config.D_MODEL)
        self.dropout = nn.Dropout(config.DROPOUT)

    def forward(self, features):
        masks = ops.zero_row_mask(features)[:, None, None, :]          # (B, 1, 1, N) bool
        return ops.linear(features, self.proj.weight, self.proj.bias), masks


@META_TEXT_EMBEDDING.register()
class UsualEmbedding(nn.Module):
    """Token embedding table; only the ``WORD_EMBEDDING: null`` form is in scope."""

    def __init__(self, config, vocab):
        super().__init__()
        if config.WORD_EMBEDDING is not None:
            raise NotImplementedError("pretrained word vectors need a network download; only "
                                      "WORD_EMBEDDING: null is supported")
        self.padding_idx = vocab.padding_idx
        self.components = nn.Embedding(len(vocab), config.D_MODEL, vocab.padding_idx)

    def forward(self, tokens, positions=None, position_table=None):
        """Returns ``(emb[tokens] (+ position_table[positions]), (padding_mask, causal_mask))``."""
        features = ops.embed(tokens, self.components.weight, positions, position_table)
        padding_masks = (tokens == self.padding_idx)[:, None, None, :]
        seq_len = tokens.shape[-1]
        sequential_masks = torch.triu(torch.ones(seq_len, seq_len, dtype=torch.bool, device=tokens.device),
                                      diagonal=1)[None, None]
        return features, (padding_masks, sequential_masks)


class SinusoidPositionalEmbedding(nn.Module):
    """DETR-style 1-D sinusoid over the region index; only the shape of ``x`` is used.

    Position of entry i is the count of unmasked entries up to i (1-based); channel c divides by
    ``temperature ** (2*floor(c/2)/num_pos_feats)``; even channels sin, odd channels cos.
    """

    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats, self.temperature, self.normalize = num_pos_feats, temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale

    def forward(self, x, mask=None):
        return ops.region_position_encoding(x.shape[0], x.shape[1], self.num_pos_feats, float(self.temperature),
                                            mask=mask, normalize=self.normalize, scale=self.scale,
                                            device=x.device)
