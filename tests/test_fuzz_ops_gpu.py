"""Seeded random sweep of the operator entry points (``openviic_amd.ops`` -> the C ABI of include/ovc.h) against fp64 torch.

The parametrised operator tests sit at the model's shapes and at hand-picked edges; this sweep draws shapes at random in between
(row counts with tails of every tile size, K not a multiple of any K tile, widths that are not multiples of the vector width where
the operator allows it, every optional operand on and off).  ``OVC_FUZZ_CASES=n`` scales every loop (default 40 draws per
operator); ``OVC_FUZZ_SEED`` moves the stream.
"""
import math
import os
import random

import numpy as np
import pytest
import torch

from openviic_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"
CASES = int(os.environ.get("OVC_FUZZ_CASES", "40"))
SEED = int(os.environ.get("OVC_FUZZ_SEED", "4102"))


def _close(got, want, tol, what):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    scale = max(want.abs().max().item(), 1e-6)
    err = (got - want).abs().max().item()
    assert err <= tol * scale + 1e-7, "{}: max abs err {:.3e} vs scale {:.3e}".format(what, err, scale)


def test_linear_random_shapes():
    """``act([x | x2] W^T + b) + residual`` (attentions.py:47, positionwise_feed_forward.py:23-28, decoders.py:59-66)."""
    rng = random.Random(SEED)
    g = torch.Generator().manual_seed(SEED)
    for case in range(CASES):
        M = rng.choice([1, 2, 5, 31, 32, 33, 63, 65, 100, 127, 129, 250, 257, 640, 1000])
        N = rng.choice([1, 3, 4, 8, 31, 32, 33, 64, 65, 96, 127, 128, 200, 512, 1000])
        K1 = 4 * rng.randint(1, 80) if rng.random() < 0.7 else 32 * rng.randint(1, 16)
        two = rng.random() < 0.3
        if two:
            K1 = 32 * rng.randint(1, 8)                        # the seam of a two-block product sits on a K tile (ovc.h)
        K2 = 4 * rng.randint(1, 40) if two else 0
        bias, relu, res = rng.random() < 0.7, rng.random() < 0.4, rng.random() < 0.4
        what = "case {}: M={} N={} K1={} K2={} bias={} relu={} residual={}".format(case, M, N, K1, K2, bias, relu, res)
        x, w = torch.randn(M, K1, generator=g), torch.randn(N, K1 + K2, generator=g) / math.sqrt(K1 + K2)
        x2 = torch.randn(M, K2, generator=g) if two else None
        b = torch.randn(N, generator=g) if bias else None
        r = torch.randn(M, N, generator=g) if res else None
        want = torch.cat([x, x2], 1).double() @ w.double().T if two else x.double() @ w.double().T
        if bias:
            want = want + b.double()
        if relu:
            want = torch.relu(want)
        if res:
            want = want + r.double()
        got = ops.linear(x.to(DEV), w.to(DEV), None if b is None else b.to(DEV), relu=relu,
                         residual=None if r is None else r.to(DEV), x2=None if x2 is None else x2.to(DEV))
        assert tuple(got.shape) == (M, N), what
        _close(got, want, 2e-5, what)


def test_attention_random_shapes():
    """Scaled dot-product attention with every optional operand (attentions.py:44-58, :97-114, :158-185): padding masks
    (per key, per query-key), geometry weights, memory slots; head sizes 4..64, up to 128 keys."""
    rng = random.Random(SEED + 1)
    g = torch.Generator().manual_seed(SEED + 1)
    for case in range(CASES):
        b, h = rng.randint(1, 4), rng.choice([1, 2, 3, 4, 8])
        dk = rng.choice([4, 8, 16, 32, 64])
        dv = dk if rng.random() < 0.8 else rng.choice([4, 8, 16, 32, 64])
        nq, nk = rng.choice([1, 2, 5, 16, 17, 50, 64, 99, 128]), rng.choice([1, 2, 5, 16, 17, 50, 64, 99, 120])
        m = rng.choice([0, 0, 1, 8]) if nk <= 120 else 0
        use_geo, mask_kind = rng.random() < 0.3 and m == 0, rng.choice(["none", "key", "full"])
        what = "case {}: b={} h={} dk={} dv={} nq={} nk={} m={} geometry={} mask={}".format(case, b, h, dk, dv, nq, nk, m, use_geo, mask_kind)
        q, k, v = torch.randn(b, nq, h * dk, generator=g), torch.randn(b, nk, h * dk, generator=g), torch.randn(b, nk, h * dv, generator=g)
        mask = None
        if mask_kind == "key":
            mask = torch.rand(b, 1, 1, nk, generator=g) < 0.3
            mask[..., 0] = False                                          # keep one key: an all-masked row is NaN in the reference
        elif mask_kind == "full":
            mask = torch.rand(b, 1, nq, nk, generator=g) < 0.3
            mask[..., 0] = False
        geo = torch.rand(b, h, nq, nk, generator=g) + 0.05 if use_geo else None
        mem = None
        if m:
            mem = (torch.randn(1, m, h * dk, generator=g), torch.randn(1, m, h * dv, generator=g), math.sqrt(dk), math.sqrt(m))
        # fp64 restatement
        qh = q.double().view(b, nq, h, dk).permute(0, 2, 1, 3)
        kk, vv = k.double(), v.double()
        if m:
            kk = torch.cat([kk, (mem[2] * mem[0].double()).expand(b, m, h * dk)], 1)
            vv = torch.cat([vv, (mem[3] * mem[1].double()).expand(b, m, h * dv)], 1)
        kh = kk.view(b, nk + m, h, dk).permute(0, 2, 3, 1)
        vh = vv.view(b, nk + m, h, dv).permute(0, 2, 1, 3)
        att = qh @ kh / math.sqrt(dk)
        if mask is not None:
            att[..., :nk] = att[..., :nk].masked_fill(mask, float("-inf"))
        if geo is not None:
            att = torch.log(torch.clamp(geo.double(), min=1e-6)) + att
        want = (torch.softmax(att, -1) @ vh).permute(0, 2, 1, 3).reshape(b, nq, h * dv)
        got = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), h, mask=None if mask is None else mask.to(DEV),
                            geometry=None if geo is None else geo.to(DEV),
                            memory=None if mem is None else (mem[0].to(DEV), mem[1].to(DEV), mem[2], mem[3]))
        assert tuple(got.shape) == (b, nq, h * dv), what
        _close(got, want, 2e-5, what)


def test_layer_norm_and_row_ops_random_shapes():
    """LayerNorm with residual / add / zero_rows (attentions.py:304-310, positionwise_feed_forward.py:26-28, encoders.py:35-40),
    log-softmax (decoders.py:123), the padding mask (models/utils.py:48-61) and the sigmoid gates."""
    rng = random.Random(SEED + 2)
    g = torch.Generator().manual_seed(SEED + 2)
    for case in range(CASES):
        rows, d = rng.choice([1, 3, 4, 7, 64, 129, 1000]), 4 * rng.randint(1, 512)
        res, zero = rng.random() < 0.5, rng.random() < 0.4
        add_rows = rng.choice([0, 1, rows]) if rows > 1 else 0
        what = "case {}: rows={} d={} residual={} add_rows={} zero_rows={}".format(case, rows, d, res, add_rows, zero)
        x = torch.randn(rows, d, generator=g) * 3 + rng.choice([0.0, 5.0, -40.0])
        gamma, beta = 1 + 0.3 * torch.randn(d, generator=g), 0.2 * torch.randn(d, generator=g)
        r = torch.randn(rows, d, generator=g) if res else None
        add = torch.randn(add_rows, d, generator=g) if add_rows else None
        zr = torch.rand(rows, generator=g) < 0.3 if zero else None
        s = x.double() + (r.double() if res else 0)
        want = torch.nn.functional.layer_norm(s, (d,), gamma.double(), beta.double(), 1e-5)
        if add_rows:
            want = want + (add.double() if add_rows == rows else add.double().expand(rows, d))
        if zero:
            want = want.masked_fill(zr.unsqueeze(-1), 0.0)
        got = ops.layer_norm(x.to(DEV), gamma.to(DEV), beta.to(DEV), residual=None if r is None else r.to(DEV),
                             add=None if add is None else add.to(DEV), zero_rows=None if zr is None else zr.to(DEV))
        _close(got, want, 1e-5, what)
        n = rng.choice([1, 2, 5, 33, 257, 1000, 10201])
        y = torch.randn(rng.choice([1, 4, 65]), n, generator=g) * 4
        _close(ops.log_softmax(y.to(DEV)), torch.log_softmax(y.double(), -1), 2e-6, what + " log_softmax n={}".format(n))
        feats = torch.randn(rows, d, generator=g)
        feats[torch.rand(rows, generator=g) < 0.3] = 0.0
        assert torch.equal(ops.zero_row_mask(feats.to(DEV)).cpu(), feats.sum(-1) == 0), what + " zero_row_mask"
        a, gate = torch.randn(rows, d, generator=g), torch.randn(rows, d, generator=g)
        _close(ops.sigmoid_gate(a.to(DEV), gate.to(DEV)), a.double() * torch.sigmoid(gate.double()), 2e-6, what + " sigmoid_gate")
