"""Operator-level Python wrappers over the HIP library (one function per ``ovc_*`` operator).

Inputs must be fp32 tensors on the HIP device; outputs are freshly allocated with ``torch.empty``
(PyTorch is only the allocator here).  Anything else raises: there is no ATen fallback for any operator.

Scope of that statement: every contraction, normalisation, softmax, selection and embedding on this step-wise
API is a HIP kernel of ``libovc.so``.  The glue AROUND the operators in ``openviic_amd/modules`` -- adding a
positional table to a combined stream, copying a level into a stacked buffer, ``alive * (prev != eos)``,
``cat`` / ``gather`` of decode state in the host-loop beam search -- is plain tensor arithmetic on the device
(ATen).  That glue is off the production path: ``beam_search(fused=True)`` runs entirely inside the engine.
"""
from typing import Optional, Tuple

import torch

from . import native
from .native import check


def _dev(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise native.OvcError("{} must live on the HIP device (got {}); openviic_amd has no CPU path"
                              .format(name, getattr(t, "device", type(t))))
    if t.dtype != dtype:
        raise native.OvcError("{} must be {} (got {})".format(name, dtype, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def _expect(t: torch.Tensor, shape, name: str) -> None:
    """The kernels index with the caller's sizes: a tensor of another shape would be read out of bounds."""
    if tuple(t.shape) != tuple(shape):
        raise native.OvcError("{} must have shape {}; got {}".format(name, tuple(shape), tuple(t.shape)))


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _as_u8(mask: torch.Tensor, name: str) -> torch.Tensor:
    if mask.dtype == torch.bool:
        mask = mask.contiguous().view(torch.uint8)
    return _dev(mask, name, torch.uint8)


def linear(x, weight, bias=None, *, relu=False, residual=None, x2=None):
    """``act([x | x2] @ weight.T + bias) + residual`` over the last dimension."""
    lib = native.load()
    x = _dev(x, "x"); weight = _dev(weight.detach(), "weight")
    lead = x.shape[:-1]
    k1 = x.shape[-1]
    x2d = x.reshape(-1, k1)
    m = x2d.shape[0]
    k2 = 0
    x2_2d = None
    if x2 is not None:
        x2 = _dev(x2, "x2")
        k2 = x2.shape[-1]
        x2_2d = x2.reshape(-1, k2)
        _expect(x2_2d, (m, k2), "x2 (flattened)")
    n = weight.shape[0]
    if weight.shape[1] != k1 + k2:
        raise native.OvcError("weight is {} but inputs have {} features".format(tuple(weight.shape), k1 + k2))
    bias = None if bias is None else _dev(bias.detach(), "bias")
    if bias is not None:
        _expect(bias, (n,), "bias")
    res2d = None
    if residual is not None:
        res2d = _dev(residual, "residual").reshape(-1, n)
        _expect(res2d, (m, n), "residual (flattened)")
    y = torch.empty(m, n, dtype=torch.float32, device=x.device)
    check(lib.ovc_linear(_ptr(x2d), k1, _ptr(x2_2d), k2, k1, k2, _ptr(weight), _ptr(bias), _ptr(res2d), n,
                         _ptr(y), n, m, n, 1 if relu else 0, native.stream_handle()), "ovc_linear")
    return y.view(*lead, n)


def layer_norm(x, gamma, beta, *, residual=None, add=None, zero_rows=None, eps=1e-5):
    """``LayerNorm(x + residual) * gamma + beta + add`` with optional row zeroing.

    ``add`` has shape ``(..., rows_a, d)`` with ``rows_a`` dividing the row count (broadcast over
    the batch); ``zero_rows`` is a bool/uint8 tensor with one entry per row.
    """
    lib = native.load()
    x = _dev(x, "x")
    d = x.shape[-1]
    rows = x.numel() // d
    residual = None if residual is None else _dev(residual, "residual")
    if residual is not None:
        _expect(residual, x.shape, "residual")
    gamma, beta = _dev(gamma.detach(), "gamma"), _dev(beta.detach(), "beta")
    _expect(gamma, (d,), "gamma")
    _expect(beta, (d,), "beta")
    add_rows = 0
    if add is not None:
        add = _dev(add, "add")
        add_rows = add.numel() // d
        if add.shape[-1] != d or add_rows == 0 or rows % add_rows:
            raise native.OvcError("add rows {} do not divide rows {}".format(add_rows, rows))
    if zero_rows is not None:
        zero_rows = _as_u8(zero_rows, "zero_rows")
        if zero_rows.numel() != rows:
            raise native.OvcError("zero_rows has {} entries for {} rows".format(zero_rows.numel(), rows))
    y = torch.empty_like(x)
    check(lib.ovc_layer_norm(_ptr(x), _ptr(residual), _ptr(gamma), _ptr(beta),
                             _ptr(add), add_rows, _ptr(zero_rows), float(eps), _ptr(y), rows, d,
                             native.stream_handle()), "ovc_layer_norm")
    return y


def attention(q, k, v, heads: int, *, mask=None, geometry=None,
              memory: Optional[Tuple[torch.Tensor, torch.Tensor, float, float]] = None):
    """Scaled dot-product attention on projected heads: q (b,nq,h*dk), k (b,nk,h*dk), v (b,nk,h*dv).

    ``mask``: bool, broadcastable ``(b|1, 1, nq|1, nk)``, True = masked.  ``geometry``:
    ``(b,h,nq,nk)`` additive-in-log weights.  ``memory``: ``(m_k, m_v, scale_k, scale_v)``.
    """
    lib = native.load()
    q, k, v = _dev(q, "q"), _dev(k, "k"), _dev(v, "v")
    b, nq, hdk = q.shape
    nk = k.shape[1]
    _expect(k, (b, nk, hdk), "k")
    if v.dim() != 3 or v.shape[:2] != (b, nk) or hdk % heads or v.shape[2] % heads:
        raise native.OvcError("v must be ({}, {}, heads*d_v) and q/k widths multiples of heads={}; got {}".format(b, nk, heads, tuple(v.shape)))
    dk, dv = hdk // heads, v.shape[2] // heads
    mask_sb = mask_sq = 0
    if mask is not None:
        if mask.dim() != 4 or mask.shape[1] != 1 or mask.shape[3] != nk:
            raise native.OvcError("mask must be (b|1, 1, nq|1, nk); got {}".format(tuple(mask.shape)))
        mask = _as_u8(mask, "mask")
        mask_sq = nk if mask.shape[2] != 1 else 0
        mask_sb = (mask.shape[2] * nk) if mask.shape[0] != 1 else 0
        if mask.shape[0] not in (1, b) or mask.shape[2] not in (1, nq):
            raise native.OvcError("mask {} does not broadcast to ({}, 1, {}, {})".format(tuple(mask.shape), b, nq, nk))
    if geometry is not None:
        geometry = _dev(geometry, "geometry")
        if tuple(geometry.shape) != (b, heads, nq, nk):
            raise native.OvcError("geometry must be {}; got {}".format((b, heads, nq, nk), tuple(geometry.shape)))
    m_k = m_v = None
    m, sk, sv = 0, 1.0, 1.0
    if memory is not None:
        m_k, m_v, sk, sv = memory
        m_k, m_v = _dev(m_k.detach(), "m_k"), _dev(m_v.detach(), "m_v")
        m = m_k.shape[-2]
        _expect(m_k.reshape(m, -1), (m, hdk), "m_k")
        _expect(m_v.reshape(m, -1), (m, heads * dv), "m_v")
    out = torch.empty(b, nq, heads * dv, dtype=torch.float32, device=q.device)
    check(lib.ovc_attention(_ptr(q), _ptr(k), _ptr(v), b, nq, nk, heads, dk, dv, _ptr(mask), mask_sb, mask_sq,
                            _ptr(geometry), _ptr(m_k), _ptr(m_v), m, float(sk), float(sv), _ptr(out),
                            native.stream_handle()), "ovc_attention")
    return out


def zero_row_mask(x):
    """bool ``(...,)``: True where the last-dimension sum of ``x`` is exactly 0."""
    lib = native.load()
    x = _dev(x, "x")
    d = x.shape[-1]
    rows = x.numel() // d
    mask = torch.empty(x.shape[:-1], dtype=torch.uint8, device=x.device)
    check(lib.ovc_zero_row_mask(_ptr(x), rows, d, _ptr(mask), native.stream_handle()), "ovc_zero_row_mask")
    return mask.view(torch.bool)


def region_position_encoding(batch: int, n: int, d: int, temperature: float = 10000.0, *, mask=None,
                             normalize=False, scale=6.283185307179586, device=None):
    lib = native.load()
    if mask is not None:
        mask = _as_u8(mask, "mask")
        _expect(mask, (batch, n), "mask")
        device = mask.device
    pe = torch.empty(batch, n, d, dtype=torch.float32, device=device)
    if not pe.is_cuda:
        raise native.OvcError("position encoding needs a HIP device")
    check(lib.ovc_region_position_encoding(_ptr(mask), batch, n, d, float(temperature), 1 if normalize else 0,
                                           float(scale), _ptr(pe), native.stream_handle()),
          "ovc_region_position_encoding")
    return pe


def embed(tokens, table, positions=None, position_table=None):
    lib = native.load()
    tokens = _dev(tokens, "tokens", torch.int64)
    table = _dev(table.detach(), "table")
    d = table.shape[1]
    pos_rows = 0
    if positions is not None:
        positions = _dev(positions.expand_as(tokens), "positions", torch.int64)
        position_table = _dev(position_table.detach(), "position_table")
        pos_rows = position_table.shape[0]
        _expect(position_table, (pos_rows, d), "position_table")
    y = torch.empty(*tokens.shape, d, dtype=torch.float32, device=tokens.device)
    check(lib.ovc_embed(_ptr(tokens), _ptr(positions), _ptr(table), table.shape[0], _ptr(position_table), pos_rows, _ptr(y),
                        tokens.numel(), d, native.stream_handle()), "ovc_embed")
    return y


def sigmoid_gate(a, g):
    lib = native.load()
    a, g = _dev(a, "a"), _dev(g, "g")
    _expect(g, a.shape, "g")
    y = torch.empty_like(a)
    check(lib.ovc_sigmoid_gate(_ptr(a), _ptr(g), _ptr(y), a.numel(), native.stream_handle()), "ovc_sigmoid_gate")
    return y


def gated_accumulate(acc, alpha, x, divisor: float = 1.0):
    lib = native.load()
    alpha, x = _dev(alpha, "alpha"), _dev(x, "x")
    acc = None if acc is None else _dev(acc, "acc")
    _expect(alpha, x.shape, "alpha")
    if acc is not None:
        _expect(acc, x.shape, "acc")
    out = torch.empty_like(x)
    check(lib.ovc_gated_accumulate(_ptr(acc), _ptr(alpha), _ptr(x), float(divisor), _ptr(out), x.numel(),
                                   native.stream_handle()), "ovc_gated_accumulate")
    return out


def log_softmax(x):
    lib = native.load()
    x = _dev(x, "x")
    n = x.shape[-1]
    y = torch.empty_like(x)
    check(lib.ovc_log_softmax(_ptr(x), _ptr(y), x.numel() // n, n, native.stream_handle()), "ovc_log_softmax")
    return y


def box_relation_weights(boxes, fc_weight, fc_bias, trignometric: bool):
    """``relu(Linear(d_g, 1))`` per head over the pairwise box geometry -> ``(B, h, N, N)``."""
    lib = native.load()
    boxes = _dev(boxes, "boxes")
    fc_weight, fc_bias = _dev(fc_weight.detach(), "fc_weight"), _dev(fc_bias.detach(), "fc_bias")
    b, n = boxes.shape[:2]
    _expect(boxes, (b, n, 4), "boxes")
    h, d_g = fc_weight.shape
    _expect(fc_bias, (h,), "fc_bias")
    w = torch.empty(b, h, n, n, dtype=torch.float32, device=boxes.device)
    check(lib.ovc_box_relation_weights(_ptr(boxes), b, n, _ptr(fc_weight), _ptr(fc_bias), h, d_g,
                                       1 if trignometric else 0, _ptr(w), native.stream_handle()),
          "ovc_box_relation_weights")
    return w


def beam_select(logp, running, alive, prev_words, eos_idx: int, k: int):
    """One selection step: returns ``(chosen (B,k) int64, score (B,k), masked logp, alive)``.

    ``alive`` comes in as the previous step's flags (B, width, 1); the ``prev != eos`` update of
    ``beam_search.py:50-51`` is applied here before the selection.
    """
    lib = native.load()
    logp = _dev(logp, "logp")
    B, width, V = logp.shape
    alive = alive[:, :width]
    if prev_words is not None:
        alive = alive * (prev_words != eos_idx).to(alive.dtype).unsqueeze(-1)
    alive_c = _dev(alive.reshape(B, width), "alive")
    running_c = _dev(running.reshape(B, -1).expand(B, width), "running")
    if not 1 <= k <= min(native.OVC_MAX_BEAM, width * V):
        raise native.OvcError("beam size {} outside 1..{}".format(k, min(native.OVC_MAX_BEAM, width * V)))
    chosen = torch.empty(B, k, dtype=torch.int64, device=logp.device)
    score = torch.empty(B, k, dtype=torch.float32, device=logp.device)
    masked = torch.empty_like(logp)
    scratch = torch.empty(8 * B * width * k, dtype=torch.uint8, device=logp.device)
    check(lib.ovc_beam_select(_ptr(logp), _ptr(running_c), _ptr(alive_c), B, width, V, k, _ptr(chosen), _ptr(score),
                              _ptr(masked), _ptr(scratch), scratch.numel(), native.stream_handle()), "ovc_beam_select")
    return chosen, score, masked, alive
