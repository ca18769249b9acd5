"""Encoder stacks, host side -- reference ``models/modules/encoders.py:11-112``.

``Encoder`` returns the last layer, ``MultilevelEncoder`` all layers stacked on dim 1 (Meshed-
Memory), ``GeometricEncoder`` adds the box-relation bias (Object-Relation Transformer).
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

    def forward(self, queries, keys, values, padding_mask, attention_mask, **kwargs):
        att = self.mhatt(queries=queries, keys=keys, values=values, padding_mask=padding_mask,
                         attention_mask=attention_mask, **kwargs)
        return self.pwff(att, zero_rows=padding_mask[:, 0, 0, :])


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
