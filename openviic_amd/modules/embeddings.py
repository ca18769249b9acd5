"""Vision / text / positional embeddings, host side.

``FeatureEmbedding``: reference ``models/modules/vision_embeddings.py:8-20``;
``DualFeatureEmbedding`` / ``GeometricDualFeatureEmbedding``: ``vision_embeddings.py:23-71``;
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


def grid_visibility_mask(boxes: torch.Tensor, grid_size: int) -> torch.Tensor:
    """bool ``(B, 1, n, g*g)``, True where a grid cell is NOT covered by the region box -- the reference's
    ``get_combine_masks`` (``models/utils.py:113-154``) without its per-box Python loops.

    A coordinate falls in the last cell whose lower edge ``i/g`` is <= it (cell 0 when none is); the edges are
    computed in float64 and compared in float32, as the reference's tensor-vs-numpy comparison does.  A box
    covers the cell rectangle [col(x_min), col(x_max)] x [row(y_min), row(y_max)]; inverted boxes cover nothing.
    """
    g = int(grid_size)
    edges = (torch.arange(g, dtype=torch.float64) / g).to(torch.float32).to(boxes.device)
    cell = ((boxes[..., None] >= edges).sum(-1) - 1).clamp_(min=0)              # (B, n, 4) last edge <= coordinate
    col0, row0, col1, row1 = cell.unbind(-1)
    index = torch.arange(g, device=boxes.device)
    cols = (index >= col0[..., None]) & (index <= col1[..., None])              # (B, n, g)
    rows = (index >= row0[..., None]) & (index <= row1[..., None])
    covered = rows[..., :, None] & cols[..., None, :]                           # (B, n, g rows, g cols)
    return ~covered.reshape(boxes.shape[0], 1, boxes.shape[1], g * g)


@META_VISION_EMBEDDING.register()
class DualFeatureEmbedding(nn.Module):
    """Separate projections of region and grid features, each with its padding mask (``vision_embeddings.py:23-43``)."""

    def __init__(self, config):
        super().__init__()
        self.region_proj = nn.Linear(config.D_REGION_FEATURE, config.D_MODEL)
        self.region_dropout = nn.Dropout(config.DROPOUT)
        self.grid_proj = nn.Linear(config.D_GRID_FEATURE, config.D_MODEL)
        self.grid_dropout = nn.Dropout(config.DROPOUT)

    def _project(self, region_features, grid_features):
        region_masks = ops.zero_row_mask(region_features)[:, None, None, :]
        grid_masks = ops.zero_row_mask(grid_features)[:, None, None, :]
        region = ops.linear(region_features, self.region_proj.weight, self.region_proj.bias)
        grid = ops.linear(grid_features, self.grid_proj.weight, self.grid_proj.bias)
        return (region, region_masks), (grid, grid_masks)

    def forward(self, region_features, grid_features):
        return self._project(region_features, grid_features)


@META_VISION_EMBEDDING.register()
class GeometricDualFeatureEmbedding(DualFeatureEmbedding):
    """``DualFeatureEmbedding`` plus the locally-constrained cross-attention masks (``vision_embeddings.py:46-71``):
    a region sees every non-padded region and the grid cells its box covers; a grid cell sees the regions
    covering it and every non-padded cell.

    The reference's version does not run (a 5-D mask is permuted as 4-D, then (B,1,1,n) and (B,1,n,g*g) masks are
    concatenated); the shapes intended by its comments are used: ``region2all (B,1,n,n+g*g)``,
    ``grid2all (B,1,g*g,n+g*g)``, key-padding masks expanded over the query dimension.
    """

    def forward(self, region_features, region_boxes, grid_features, grid_boxes):
        (region, region_masks), (grid, grid_masks) = self._project(region_features, grid_features)
        bsz, n, gg = region.shape[0], region.shape[1], grid.shape[1]
        grid_size = int(gg ** 0.5)
        region2grid = grid_visibility_mask(region_boxes, grid_size)
        region2all = torch.cat([region_masks.expand(bsz, 1, n, n), region2grid], dim=-1)
        grid2all = torch.cat([region2grid.transpose(2, 3), grid_masks.expand(bsz, 1, gg, gg)], dim=-1)
        return (region, region_masks), (grid, grid_masks), (region2all, grid2all)


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
