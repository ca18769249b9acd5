"""Decoder stacks, host side -- reference ``models/modules/decoders.py:13-173``.

These modules implement the operator-by-operator API (teacher-forced ``forward`` and the
step-wise stateful mode).  ``BaseTransformer.beam_search`` does not go through them: it hands the
same parameters to the fused HIP engine.
"""
import torch
from torch import nn

from .. import ops
from ..builders.decoder_builder import META_DECODER
from ..builders.text_embedding_builder import build_text_embedding
from .attentions import MultiHeadAttention
from .containers import Module, ModuleList
from .feed_forward import PositionWiseFeedForward


def sinusoid_encoding_table(max_len: int, d_model: int, padding_idx=None) -> torch.Tensor:
    """Token position table (``models/utils.py:21-40``): ``[:, ::2] = sin``, ``[:, 1::2] = cos`` of
    ``pos / 10000**(2i/d)`` for i < d/2; the padding row is zero."""
    pos = torch.arange(max_len, dtype=torch.float32).view(-1, 1)
    dim = torch.arange(d_model // 2, dtype=torch.float32).view(1, -1)
    table = torch.zeros(max_len, d_model)
    table[:, ::2] = torch.sin(pos / 10000 ** (2 * dim / d_model))
    table[:, 1::2] = torch.cos(pos / 10000 ** (2 * dim / d_model))
    if padding_idx is not None:
        table[padding_idx] = 0
    return table


class DecoderLayer(Module):
    """self-attention -> cross-attention -> feed-forward; rows fed a <pad> token are cleared."""

    def __init__(self, config):
        super().__init__()
        self.self_attn = MultiHeadAttention(config.SELF_ATTENTION)
        self.enc_attn = MultiHeadAttention(config.ENC_ATTENTION)
        self.pwff = PositionWiseFeedForward(config.ENC_ATTENTION)

    def forward(self, queries, keys, values, self_padding_mask, self_attention_mask, enc_attention_mask, **kwargs):
        self_att = self.self_attn(queries, queries, queries, padding_mask=self_padding_mask,
                                  attention_mask=self_attention_mask, **kwargs)
        enc_att = self.enc_attn(self_att, keys, values, padding_mask=self_padding_mask,
                                attention_mask=enc_attention_mask, **kwargs)
        return self.pwff(enc_att, zero_rows=self_padding_mask[:, 0, 0, :])


class MeshedDecoderLayer(Module):
    """Cross-attends to every encoder level with one shared ``enc_attn`` and mixes the results with
    sigmoid gates: ``sum_i sigmoid(W_i [self; enc_i]) * enc_i / sqrt(levels)`` (``decoders.py:51-73``)."""

    def __init__(self, config):
        super().__init__()
        self.self_attn = MultiHeadAttention(config.SELF_ATTENTION)
        self.enc_attn = MultiHeadAttention(config.ENC_ATTENTION)
        self.pwff = PositionWiseFeedForward(config.ENC_ATTENTION)
        self.fc_alphas = nn.ModuleList([nn.Linear(2 * config.D_MODEL, config.D_MODEL)
                                        for _ in range(config.N_ENCODER_LAYERS)])
        self.nlayers = config.N_ENCODER_LAYERS
        self.init_weights()

    def init_weights(self):
        for fc in self.fc_alphas:
            nn.init.xavier_uniform_(fc.weight)
            nn.init.constant_(fc.bias, 0)

    def forward(self, queries, keys, values, self_padding_mask, self_attention_mask, enc_attention_mask, **kwargs):
        self_att = self.self_attn(queries, queries, queries, padding_mask=self_padding_mask,
                                  attention_mask=self_attention_mask, **kwargs)
        mixed = None
        divisor = self.nlayers ** 0.5
        for ith, fc_alpha in enumerate(self.fc_alphas):
            enc_att = self.enc_attn(self_att, keys[:, ith], values[:, ith], padding_mask=self_padding_mask,
                                    attention_mask=enc_attention_mask, **kwargs)
            alpha = ops.linear(self_att, fc_alpha.weight, fc_alpha.bias, x2=enc_att)
            mixed = ops.gated_accumulate(mixed, alpha, enc_att, divisor if ith == self.nlayers - 1 else 1.0)
        return self.pwff(mixed, zero_rows=self_padding_mask[:, 0, 0, :])


class _DecoderBase(Module):
    layer_class = DecoderLayer

    def __init__(self, config, vocab):
        super().__init__()
        self.d_model = config.D_MODEL
        self.max_len = vocab.max_caption_length
        self.padding_idx = vocab.padding_idx
        self.N = config.LAYERS
        self.word_emb = build_text_embedding(config.TEXT_EMBEDDING, vocab)
        self.pos_emb = nn.Embedding.from_pretrained(
            sinusoid_encoding_table(self.max_len + 1, config.D_MODEL, padding_idx=0), freeze=True)
        self.layers = ModuleList([self.layer_class(config.ATTENTION) for _ in range(config.LAYERS)])
        self.fc = nn.Linear(config.D_MODEL, len(vocab), bias=False)
        self.register_state("running_mask_self_attention", torch.zeros((1, 1, 0)).bool())
        self.register_state("running_seq", torch.zeros((1,)).long())

    def forward(self, caption_tokens, encoder_features, encoder_attention_mask):
        """Log-probabilities ``(b, T, V)`` (``decoders.py:95-123``)."""
        b_s, seq_len = caption_tokens.shape[:2]
        seq = torch.arange(1, seq_len + 1, device=caption_tokens.device).view(1, -1).expand(b_s, -1)
        seq = seq.masked_fill(caption_tokens == self.padding_idx, 0)
        if self._is_stateful:
            self.running_seq.add_(1)
            seq = self.running_seq
        out, (padding_masks, sequential_masks) = self.word_emb(caption_tokens, positions=seq,
                                                               position_table=self.pos_emb.weight)
        attention_masks = torch.logical_or(padding_masks, sequential_masks)
        if self._is_stateful:
            self.running_mask_self_attention = torch.cat([self.running_mask_self_attention, attention_masks], -1)
            attention_masks = self.running_mask_self_attention
        for layer in self.layers:
            out = layer(queries=out, keys=encoder_features, values=encoder_features,
                        self_padding_mask=padding_masks, self_attention_mask=attention_masks,
                        enc_attention_mask=encoder_attention_mask)
        return ops.log_softmax(ops.linear(out, self.fc.weight, None))


@META_DECODER.register()
class Decoder(_DecoderBase):
    layer_class = DecoderLayer


@META_DECODER.register()
class MeshedDecoder(_DecoderBase):
    layer_class = MeshedDecoderLayer
