"""Encoder stacks, host side -- reference ``models/modules/encoders.py:11-211``.

``Encoder`` returns the last layer, ``MultilevelEncoder`` all layers stacked on dim 1 (Meshed-
Memory), ``GeometricEncoder`` adds the box-relation bias (Object-Relation Transformer),
``DualCollaborativeLevelEncoder`` runs region and grid streams with locally-constrained cross-attention (DLCT).
"""
import copy

import torch
from torch import nn

from .. import ops
from ..builders.encoder_builder import META_ENCODER
from .attentions import MultiHeadAttention
from .embeddings import SinusoidPositionalEmbedding
from .feed_forward import PositionWiseFeedForward


class EncoderLayer(nn.Module):
    """Multi-head attention, feed-forward, then padded query rows are cleared (``encoders.py:17-22``)."""

    def __init__(self, config):
        super().__init__()
        self.mhatt = MultiHeadAttention(config)
        self.pwff = PositionWiseFeedForward(config)

    def forward(self, queries, keys, values, padding_mask, attention_mask, query_padding_mask=None, **kwargs):
        """``query_padding_mask`` (B,1,1,nq) names the rows to clear when ``padding_mask`` is a per-query
        (B,1,nq,nk) mask (cross-attention between streams), where the reference's squeeze cannot broadcast."""
        att = self.mhatt(queries=queries, keys=keys, values=values, padding_mask=padding_mask,
                         attention_mask=attention_mask, **kwargs)
        rows = padding_mask if query_padding_mask is None else query_padding_mask
        return self.pwff(att, zero_rows=rows[:, 0, 0, :])


class _EncoderBase(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.pos_embedding = SinusoidPositionalEmbedding(config.D_MODEL)
        self.layer_norm = nn.LayerNorm(config.D_MODEL)
        self.d_model = config.D_MODEL
        self.layers = nn.ModuleList([EncoderLayer(config.SELF_ATTENTION) for _ in range(config.LAYERS)])

    def _prologue(self, features):
        """``LayerNorm(features) + PE`` (``encoders.py:36``)."""
        return ops.layer_norm(features, self.layer_norm.weight, self.layer_norm.bias,
                              add=self.pos_embedding(features), eps=self.layer_norm.eps)


@META_ENCODER.register()
class Encoder(_EncoderBase):
    def forward(self, features: torch.Tensor, padding_mask: torch.Tensor):
        out = self._prologue(features)
        for layer in self.layers:
            out = layer(queries=out, keys=out, values=out, padding_mask=padding_mask, attention_mask=padding_mask)
        return out


@META_ENCODER.register()
class MultilevelEncoder(_EncoderBase):
    def forward(self, features: torch.Tensor, padding_mask: torch.Tensor):
        out = self._prologue(features)
        levels = torch.empty(features.shape[0], len(self.layers), *features.shape[1:],
                             dtype=features.dtype, device=features.device)
        for i, layer in enumerate(self.layers):
            out = layer(queries=out, keys=out, values=out, padding_mask=padding_mask, attention_mask=padding_mask)
            levels[:, i].copy_(out)
        return levels


@META_ENCODER.register()
class GeometricEncoder(_EncoderBase):
    """Box geometry -> per-head ``relu(Linear(d_g, 1))`` weights, computed once and shared by all
    layers (``encoders.py:93-112``, ``models/utils.py:156-216``)."""

    def __init__(self, config):
        super().__init__(config)
        self.trignometric_embedding = config.TRIGNOMETRIC_EMBEDDING
        heads = config.SELF_ATTENTION.HEAD
        self.d_g = config.D_MODEL // heads if self.trignometric_embedding else 4
        self.fc_gs = nn.ModuleList([copy.deepcopy(nn.Linear(self.d_g, 1)) for _ in range(heads)])
        self.init_weights()

    def init_weights(self):
        for fc_g in self.fc_gs:
            nn.init.xavier_uniform_(fc_g.weight)
            nn.init.constant_(fc_g.bias, 0)

    def geometry_weights(self, boxes: torch.Tensor) -> torch.Tensor:
        weight = torch.cat([fc.weight for fc in self.fc_gs], dim=0)       # (h, d_g)
        bias = torch.cat([fc.bias for fc in self.fc_gs], dim=0)           # (h,)
        return ops.box_relation_weights(boxes, weight, bias, self.trignometric_embedding)

    def forward(self, features: torch.Tensor, boxes: torch.Tensor, padding_mask: torch.Tensor):
        relative_geometry_weights = self.geometry_weights(boxes)
        out = self._prologue(features)
        for layer in self.layers:
            out = layer(queries=out, keys=out, values=out, relative_geometry_weights=relative_geometry_weights,
                        padding_mask=padding_mask, attention_mask=padding_mask)
        return out


@META_ENCODER.register()
class DualCollaborativeLevelEncoder(nn.Module):
    """Region stream and grid stream, per layer: geometry self-attention inside each stream, then each stream
    attends over ``[regions ; grids] + PE`` under its visibility mask (``encoders.py:115-211``).

    One set of per-head geometry weights over the n + g*g boxes is computed once and sliced per attention.
    The reference class fails in its cross layers (its ``EncoderLayer`` clears rows with
    ``padding_mask.squeeze(1).squeeze(1)``, which cannot broadcast for a (B,1,nq,nk) mask); here those rows are
    the query stream's padded rows.  Pinned by ``tests/golden/g8_dlct_encoder*.npz`` (reference sub-modules
    composed with that repair).
    """

    def __init__(self, config):
        super().__init__()
        self.d_model = config.D_MODEL
        self.trignometric_embedding = config.TRIGNOMETRIC_EMBEDDING
        self.d_g = config.D_MODEL // config.HEAD if self.trignometric_embedding else 4
        self.layer_norm_region = nn.LayerNorm(self.d_model)
        self.layer_norm_grid = nn.LayerNorm(self.d_model)
        self.fc_gs = nn.ModuleList([nn.Linear(self.d_g, 1) for _ in range(config.HEAD)])
        self.pos_embedding = SinusoidPositionalEmbedding(config.D_MODEL, normalize=True)
        self.layers_region = nn.ModuleList([EncoderLayer(config.SELF_ATTENTION) for _ in range(config.LAYERS)])
        self.layers_grid = nn.ModuleList([EncoderLayer(config.SELF_ATTENTION) for _ in range(config.LAYERS)])
        self.region2grid = nn.ModuleList([EncoderLayer(config.CROSS_ATTENTION) for _ in range(config.LAYERS)])
        self.grid2region = nn.ModuleList([EncoderLayer(config.CROSS_ATTENTION) for _ in range(config.LAYERS)])
        self.init_weights()

    def init_weights(self):
        for fc_g in self.fc_gs:
            nn.init.xavier_uniform_(fc_g.weight)
            nn.init.constant_(fc_g.bias, 0)

    def geometry_weights(self, boxes: torch.Tensor) -> torch.Tensor:
        weight = torch.cat([fc.weight for fc in self.fc_gs], dim=0)
        bias = torch.cat([fc.bias for fc in self.fc_gs], dim=0)
        return ops.box_relation_weights(boxes, weight, bias, self.trignometric_embedding)

    def forward(self, region_features, region_boxes, region_padding_mask, region2all_mask,
                grid_features, grid_boxes, grid_padding_mask, grid2all_mask):
        n = region_features.shape[1]
        w = self.geometry_weights(torch.cat([region_boxes, grid_boxes], dim=1))          # (B, h, n+gg, n+gg)
        w_rr, w_gg = w[:, :, :n, :n].contiguous(), w[:, :, n:, n:].contiguous()
        w_ra, w_ga = w[:, :, :n, :].contiguous(), w[:, :, n:, :].contiguous()
        region = ops.layer_norm(region_features, self.layer_norm_region.weight, self.layer_norm_region.bias,
                                add=self.pos_embedding(region_features), eps=self.layer_norm_region.eps)
        grid = ops.layer_norm(grid_features, self.layer_norm_grid.weight, self.layer_norm_grid.bias,
                              add=self.pos_embedding(grid_features), eps=self.layer_norm_grid.eps)
        pe_all = None
        for l_region, l_grid, l_r2g, l_g2r in zip(self.layers_region, self.layers_grid, self.region2grid, self.grid2region):
            region = l_region(queries=region, keys=region, values=region, relative_geometry_weights=w_rr,
                              padding_mask=region_padding_mask, attention_mask=region_padding_mask)
            grid = l_grid(queries=grid, keys=grid, values=grid, relative_geometry_weights=w_gg,
                          padding_mask=grid_padding_mask, attention_mask=grid_padding_mask)
            combined = torch.cat([region, grid], dim=1)
            if pe_all is None:
                pe_all = self.pos_embedding(combined)
            combined = combined + pe_all
            region = l_r2g(queries=region, keys=combined, values=combined, relative_geometry_weights=w_ra,
                           padding_mask=region2all_mask, attention_mask=region2all_mask,
                           query_padding_mask=region_padding_mask)
            grid = l_g2r(queries=grid, keys=combined, values=combined, relative_geometry_weights=w_ga,
                         padding_mask=grid2all_mask, attention_mask=grid2all_mask,
                         query_padding_mask=grid_padding_mask)
        return torch.cat([region, grid], dim=1), torch.cat([region_padding_mask, grid_padding_mask], dim=-1)
