"""Attention operators and the multi-head wrapper, host side.

Parameter names, shapes and initialisation follow the reference
(``models/modules/attentions.py:9-58`` plain, ``:61-114`` geometry, ``:117-185`` memory,
``:270-318`` multi-head wrapper) so that reference checkpoints load key-for-key.  The arithmetic
runs in the HIP library: projections through the fp32 MFMA GEMM (``ovc_linear``), the
scale / mask / softmax / weighted-sum through ``ovc_attention``; no operator here falls back to ATen
(shape glue such as ``view`` / ``cat`` of cached state is ordinary tensor code -- see ``ops.py``).
"""
import math

import torch
from torch import nn

from .. import ops
from ..builders.attention_builder import META_ATTENTION, build_attention
from .containers import Module


class _ProjectedAttention(nn.Module):
    """fc_q / fc_k / fc_v / fc_o with Xavier-uniform weights and zero bias."""

    def __init__(self, config):
        super().__init__()
        self.d_model, self.h = config.D_MODEL, config.HEAD
        self.d_k, self.d_v = config.D_KEY, config.D_VALUE
        self.fc_q = nn.Linear(self.d_model, self.h * self.d_k)
        self.fc_k = nn.Linear(self.d_model, self.h * self.d_k)
        self.fc_v = nn.Linear(self.d_model, self.h * self.d_v)
        self.fc_o = nn.Linear(self.h * self.d_v, self.d_model)
        self.init_weights()

    def init_weights(self):
        for fc in (self.fc_q, self.fc_k, self.fc_v, self.fc_o):
            nn.init.xavier_uniform_(fc.weight)
            nn.init.constant_(fc.bias, 0)

    def _project(self, queries, keys, values):
        q = ops.linear(queries, self.fc_q.weight, self.fc_q.bias)
        k = ops.linear(keys, self.fc_k.weight, self.fc_k.bias)
        v = ops.linear(values, self.fc_v.weight, self.fc_v.bias)
        return q, k, v


@META_ATTENTION.register()
class ScaledDotProductAttention(_ProjectedAttention):
    """softmax(q k^T / sqrt(d_k) masked) v -- reference ``attentions.py:44-58``."""

    def forward(self, queries, keys, values, attention_mask=None):
        q, k, v = self._project(queries, keys, values)
        out = ops.attention(q, k, v, self.h, mask=attention_mask)
        return ops.linear(out, self.fc_o.weight, self.fc_o.bias)


@META_ATTENTION.register()
class AugmentedGeometryScaledDotProductAttention(_ProjectedAttention):
    """softmax(log(clamp(w_g, 1e-6)) + q k^T / sqrt(d_k)) v -- reference ``attentions.py:97-114``."""

    def forward(self, queries, keys, values, relative_geometry_weights, attention_mask=None):
        q, k, v = self._project(queries, keys, values)
        out = ops.attention(q, k, v, self.h, mask=attention_mask, geometry=relative_geometry_weights)
        return ops.linear(out, self.fc_o.weight, self.fc_o.bias)


@META_ATTENTION.register()
class AugmentedMemoryScaledDotProductAttention(_ProjectedAttention):
    """Attention over the keys plus ``m`` learned memory slots -- reference ``attentions.py:158-185``.

    Keys are ``[fc_k(x) ; sqrt(d_k) m_k]`` and values ``[fc_v(x) ; sqrt(m) m_v]``; the mask covers
    the real keys only.
    """

    def __init__(self, config):
        self.m = config.MEMORY
        super().__init__(config)

    def init_weights(self):
        if not hasattr(self, "m_k"):
            self.m_k = nn.Parameter(torch.empty(1, self.m, self.h * self.d_k))
            self.m_v = nn.Parameter(torch.empty(1, self.m, self.h * self.d_v))
        super().init_weights()
        nn.init.normal_(self.m_k, 0, 1 / self.d_k)
        nn.init.normal_(self.m_v, 0, 1 / self.m)

    def forward(self, queries, keys, values, attention_mask=None):
        q, k, v = self._project(queries, keys, values)
        out = ops.attention(q, k, v, self.h, mask=attention_mask,
                            memory=(self.m_k, self.m_v, math.sqrt(self.d_k), math.sqrt(self.m)))
        return ops.linear(out, self.fc_o.weight, self.fc_o.bias)


class MultiHeadAttention(Module):
    """Attention + residual LayerNorm (+ Attention-on-Attention gate) -- ``attentions.py:270-318``.

    In stateful mode the layer appends its *un-projected* inputs to ``running_keys`` /
    ``running_values`` and attends over the whole history, exactly as the reference does.  (The
    fused beam search caches projected keys/values instead; see ``csrc/engine.hip``.)
    """

    def __init__(self, config):
        super().__init__()
        d_model = config.D_MODEL
        self.use_aoa = config.USE_AOA
        if self.use_aoa:
            self.informative_attention = nn.Linear(2 * d_model, d_model)
            self.gated_attention = nn.Linear(2 * d_model, d_model)
        self.attention = build_attention(config)
        self.dropout = nn.Dropout(p=config.DROPOUT)       # identity: inference only
        self.layer_norm = nn.LayerNorm(d_model)
        self.can_be_stateful = config.CAN_BE_STATEFUL
        if self.can_be_stateful:
            self.register_state("running_keys", torch.zeros((0, d_model)))
            self.register_state("running_values", torch.zeros((0, d_model)))

    def forward(self, queries, keys, values, padding_mask, attention_mask, **kwargs):
        if self.can_be_stateful and self._is_stateful:
            self.running_keys = torch.cat([self.running_keys, keys], 1)
            self.running_values = torch.cat([self.running_values, values], 1)
            keys, values = self.running_keys, self.running_values
        out = self.attention(queries, keys, values, attention_mask=attention_mask, **kwargs)
        out = ops.layer_norm(out, self.layer_norm.weight, self.layer_norm.bias, residual=queries,
                             eps=self.layer_norm.eps)
        if self.use_aoa:
            info = ops.linear(queries, self.informative_attention.weight, self.informative_attention.bias, x2=out)
            gate = ops.linear(queries, self.gated_attention.weight, self.gated_attention.bias, x2=out)
            out = ops.sigmoid_gate(info, gate)
        return out
