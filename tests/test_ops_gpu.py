"""Operator-level parity of the HIP kernels (through the C ABI) against CPU references.

Floating-point tolerance: the north-star bar is 1e-3 relative on logits; operators are held to a
much tighter 2e-5 relative to the result scale (fp32 MFMA accumulates in fp32 with a different
summation order than the CPU BLAS).
"""
import math

import numpy as np
import pytest
import torch

from helpers import golden
from oracle.captioner import (OracleCaptioner, box_relation_features, region_position_encoding)

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _close(got, want, tol=2e-5, what=""):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    scale = max(want.abs().max().item(), 1e-6)
    err = (got - want).abs().max().item()
    assert err <= tol * scale + 1e-7, "{}: max abs err {:.3e} vs scale {:.3e}".format(what, err, scale)


@pytest.mark.parametrize("M,N,K", [(70, 53, 64), (1, 64, 32), (1280, 512, 512), (300, 10201, 512), (1000, 512, 2048),
                                   (129, 1536, 128), (257, 2048, 512), (64, 64, 4)])
@pytest.mark.parametrize("mode", ["plain", "bias_relu", "residual"])
def test_linear(M, N, K, mode):
    from openviic_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    x, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g) if mode != "plain" else None
    r = torch.randn(M, N, generator=g) if mode == "residual" else None
    want = x.double() @ w.double().T
    if b is not None:
        want = want + b.double()
    if mode == "bias_relu":
        want = want.relu()
    if r is not None:
        want = want + r.double()
    got = ops.linear(x.to(DEV), w.to(DEV), None if b is None else b.to(DEV), relu=mode == "bias_relu",
                     residual=None if r is None else r.to(DEV))
    _close(got, want, what="linear %s" % mode)


def test_linear_concatenated_input():
    from openviic_amd import ops
    g = torch.Generator().manual_seed(5)
    a, b2 = torch.randn(3, 37, 64, generator=g), torch.randn(3, 37, 64, generator=g)
    w, bias = torch.randn(64, 128, generator=g) / 11, torch.randn(64, generator=g)
    want = torch.cat([a, b2], -1).double() @ w.double().T + bias.double()
    got = ops.linear(a.to(DEV), w.to(DEV), bias.to(DEV), x2=b2.to(DEV))
    assert got.shape == (3, 37, 64)
    _close(got, want, what="linear x2")


def test_linear_rejects_cpu_tensors():
    from openviic_amd import native, ops
    with pytest.raises(native.OvcError):
        ops.linear(torch.randn(4, 8), torch.randn(8, 8))


@pytest.mark.parametrize("d", [4, 64, 192, 260, 512, 1000, 2048])       # 1, 2, 4 and 8 vectors per lane, ragged last vector
def test_layer_norm(d):
    from openviic_amd import ops
    g = torch.Generator().manual_seed(d)
    B, N = 5, 13
    x, r = torch.randn(B, N, d, generator=g) * 3 + 1, torch.randn(B, N, d, generator=g)
    gamma, beta = torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g)
    add = torch.randn(1, N, d, generator=g)
    zero = torch.rand(B, N, generator=g) < 0.3
    want = torch.nn.functional.layer_norm((x + r).double(), (d,), gamma.double(), beta.double()) + add.double()
    want = want.masked_fill(zero[..., None], 0)
    got = ops.layer_norm(x.to(DEV), gamma.to(DEV), beta.to(DEV), residual=r.to(DEV), add=add.to(DEV), zero_rows=zero.to(DEV))
    _close(got, want, tol=1e-5, what="layer_norm")
    plain = ops.layer_norm(x.to(DEV), gamma.to(DEV), beta.to(DEV))
    _close(plain, torch.nn.functional.layer_norm(x.double(), (d,), gamma.double(), beta.double()), tol=1e-5, what="ln plain")


def _sdpa_ref(q, k, v, h, mask=None, geometry=None, memory=None):
    b, nq, _ = q.shape
    dk, dv = q.shape[2] // h, v.shape[2] // h
    nk = k.shape[1]
    if memory is not None:
        m_k, m_v, sk, sv = memory
        k = torch.cat([k, sk * m_k.expand(b, -1, -1)], 1)
        v = torch.cat([v, sv * m_v.expand(b, -1, -1)], 1)
    qh = q.double().view(b, nq, h, dk).permute(0, 2, 1, 3)
    kh = k.double().view(b, -1, h, dk).permute(0, 2, 3, 1)
    vh = v.double().view(b, -1, h, dv).permute(0, 2, 1, 3)
    att = qh @ kh / math.sqrt(dk)
    if mask is not None:
        att[..., :nk] = att[..., :nk].masked_fill(mask, float("-inf"))
    if geometry is not None:
        att = torch.log(torch.clamp(geometry.double(), min=1e-6)) + att
    return (torch.softmax(att, -1) @ vh).permute(0, 2, 1, 3).reshape(b, nq, h * dv)


@pytest.mark.parametrize("b,nq,nk,h,dk", [(3, 50, 50, 8, 64), (2, 7, 7, 4, 16), (2, 20, 20, 8, 64), (2, 50, 99, 8, 64),
                                          (1, 70, 33, 2, 32), (2, 128, 128, 1, 64),
                                          # beyond 128 keys / queries (round 4): the key-tiled kernel (online softmax over 128-key
                                          # tiles) and, for nq > 128 with few keys, the LDS-score kernel over 64-query tiles
                                          (2, 100, 100, 8, 64), (2, 50, 246, 8, 64), (2, 196, 196, 8, 64), (1, 300, 257, 2, 32),
                                          (1, 129, 129, 3, 16), (2, 130, 60, 4, 16), (1, 5, 1000, 2, 64)])
@pytest.mark.parametrize("kind", ["keymask", "querymask", "geometry", "memory", "nomask"])
def test_attention(b, nq, nk, h, dk, kind):
    from openviic_amd import ops
    g = torch.Generator().manual_seed(nq * 31 + nk + h)
    q, k, v = (torch.randn(b, n, h * dk, generator=g) for n in (nq, nk, nk))
    mask = geometry = memory = None
    if kind == "keymask":
        mask = torch.rand(b, 1, 1, nk, generator=g) < 0.3
        mask[..., 0] = False
    elif kind in ("querymask", "geometry"):
        mask = torch.rand(b, 1, nq, nk, generator=g) < 0.3
        mask[..., 0] = False
        if kind == "geometry":
            geometry = torch.rand(b, h, nq, nk, generator=g) * 2 - 0.5
    elif kind == "memory":
        m = 40                      # meshed_memory_transformer.yaml; with 100 regions: 140 keys
        memory = (torch.randn(1, m, h * dk, generator=g) / dk, torch.randn(1, m, h * dk, generator=g) / m,
                  math.sqrt(dk), math.sqrt(m))
        mask = torch.rand(b, 1, 1, nk, generator=g) < 0.3
        mask[..., 0] = False
    want = _sdpa_ref(q, k, v, h, mask, geometry, memory)
    mem_dev = None if memory is None else (memory[0].to(DEV), memory[1].to(DEV), memory[2], memory[3])
    got = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), h, mask=None if mask is None else mask.to(DEV),
                        geometry=None if geometry is None else geometry.to(DEV), memory=mem_dev)
    _close(got, want, what="attention %s" % kind)


def test_attention_key_tiles_that_are_entirely_masked():
    """The key-tiled kernel's online softmax must pass over tiles in which a query sees nothing (running maximum still -inf:
    no rescale by exp(-inf - -inf)), over queries whose FIRST visible key comes late, and must give NaN -- as torch.softmax
    over a row of -inf does -- for a query that sees no key at all."""
    from openviic_amd import ops
    g = torch.Generator().manual_seed(77)
    b, nq, nk, h, dk = 2, 140, 400, 4, 32
    q, k, v = (torch.randn(b, n, h * dk, generator=g) for n in (nq, nk, nk))
    mask = torch.rand(b, 1, nq, nk, generator=g) < 0.2
    mask[0, 0, 3, :256] = True              # two whole tiles hidden, then visible keys
    mask[0, 0, 4, 128:384] = True           # the middle tiles hidden
    mask[0, 0, 5, 130:] = True              # nothing after the second tile's first keys
    mask[1, 0, 7, :] = True                 # no key at all -> NaN
    mask[1, 0, 8, :399] = True              # only the very last key
    mask[1, 0, 8, 399] = False
    want = _sdpa_ref(q, k, v, h, mask)
    got = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), h, mask=mask.to(DEV)).cpu()
    assert torch.isnan(want[1, 7]).all() and torch.isnan(got[1, 7]).all()
    keep = torch.ones(b, nq, dtype=torch.bool)
    keep[1, 7] = False
    assert torch.isfinite(got[keep]).all()
    _close(got[keep], want[keep], what="masked key tiles")
    np.testing.assert_allclose(got[1, 8].numpy(), v[1, 399].numpy(), rtol=0, atol=1e-6)    # one visible key: its value row
    # key-padding form with memory slots: the slots are never masked, so even an image without regions is finite there
    keymask = torch.rand(b, 1, 1, nk, generator=g) < 0.3
    keymask[0] = True
    m = 40
    memory = (torch.randn(1, m, h * dk, generator=g) / dk, torch.randn(1, m, h * dk, generator=g) / m, math.sqrt(dk), math.sqrt(m))
    got = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), h, mask=keymask.to(DEV),
                        memory=(memory[0].to(DEV), memory[1].to(DEV), memory[2], memory[3]))
    _close(got, _sdpa_ref(q, k, v, h, keymask, None, memory), what="memory slots behind masked tiles")


def test_attention_general_kernel_still_agrees():
    """The register-resident kernel serves up to 128 queries x 128 keys, the key-tiled one everything beyond 128 keys; more than
    128 queries over at most 128 keys take the LDS-score kernel (64-query tiles).  All three against fp64 on one problem
    family: memory slots + key mask and geometry + per-query mask.  (Until round 3 an environment switch forced the LDS-score
    kernel; the shipped library no longer reads such switches -- csrc/common.h -- so the shapes select the kernels here.)"""
    from openviic_amd import ops
    g = torch.Generator().manual_seed(12)
    for nq, nk in ((50, 50), (150, 50), (150, 120)):            # registers / LDS scores / key tiles (120 + 40 slots)
        b, h, dk, m = 2, 8, 64, 40
        q, k, v = (torch.randn(b, n, h * dk, generator=g) for n in (nq, nk, nk))
        keymask = torch.rand(b, 1, 1, nk, generator=g) < 0.3
        keymask[..., 0] = False
        qmask = torch.rand(b, 1, nq, nk, generator=g) < 0.3
        qmask[..., 0] = False
        geometry = torch.rand(b, h, nq, nk, generator=g) * 2 - 0.5
        memory = (torch.randn(1, m, h * dk, generator=g) / dk, torch.randn(1, m, h * dk, generator=g) / m, math.sqrt(dk), math.sqrt(m))
        mem_dev = (memory[0].to(DEV), memory[1].to(DEV), memory[2], memory[3])
        got = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), h, mask=keymask.to(DEV), memory=mem_dev)
        _close(got, _sdpa_ref(q, k, v, h, keymask, None, memory), what="memory, nq=%d nk=%d" % (nq, nk))
        got = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), h, mask=qmask.to(DEV), geometry=geometry.to(DEV))
        _close(got, _sdpa_ref(q, k, v, h, qmask, geometry), what="geometry, nq=%d nk=%d" % (nq, nk))


def test_causal_mask_attention_matches_teacher_forcing_shape():
    from openviic_amd import ops
    g = torch.Generator().manual_seed(3)
    b, T, h, dk = 4, 20, 8, 64
    q, k, v = (torch.randn(b, T, h * dk, generator=g) for _ in range(3))
    pad = torch.zeros(b, 1, 1, T, dtype=torch.bool)
    pad[1, ..., 5] = True
    mask = pad | torch.triu(torch.ones(T, T), diagonal=1).bool()[None, None]
    got = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), h, mask=mask.to(DEV))
    _close(got, _sdpa_ref(q, k, v, h, mask), what="causal attention")


def test_dlct_cross_attention_module_against_reference_golden():
    """G4: reference MultiHeadAttention(AugmentedGeometry...), nq=50, nk=99, per-query mask."""
    from openviic_amd.config import ConfigNode
    from openviic_amd.modules import MultiHeadAttention
    from openviic_amd.utils.synthetic import synthetic_state_dict
    g = golden("g4_dlct_cross_attention.npz")
    att = ConfigNode(dict(ARCHITECTURE="AugmentedGeometryScaledDotProductAttention", HEAD=8, D_MODEL=512, D_KEY=64,
                          D_VALUE=64, D_FF=2048, USE_AOA=False, CAN_BE_STATEFUL=False, DROPOUT=0.1))
    mha = MultiHeadAttention(att).eval()
    sd = synthetic_state_dict({"x." + k: v for k, v in mha.state_dict().items()}, seed=31, mode="generic")
    mha.load_state_dict({k[2:]: v for k, v in sd.items()})
    mha.to(DEV)
    t = lambda name: torch.from_numpy(g[name]).to(DEV)
    out = mha(queries=t("queries"), keys=t("keys"), values=t("keys"), padding_mask=None, attention_mask=t("mask"),
              relative_geometry_weights=t("geometry"))
    _close(out, torch.from_numpy(g["out"]), tol=5e-5, what="DLCT cross attention")


def test_zero_row_mask_and_position_encoding():
    from openviic_amd import ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(6, 50, 2048, generator=g)
    x[0, 40:] = 0
    x[3, 7] = 0
    x[5, 2, :] = 0
    x[5, 2, 3], x[5, 2, 4] = 1.5, -1.5           # cancels to exactly 0 -> counts as padding in the reference too
    got = ops.zero_row_mask(x.to(DEV))
    np.testing.assert_array_equal(got.cpu().numpy(), (x.sum(-1) == 0).numpy())
    pe = ops.region_position_encoding(2, 50, 512, device=DEV)
    _close(pe, region_position_encoding(2, 50, 512), tol=1e-5, what="region PE")  # powf vs torch.pow: 1 ulp on the divisor x angle<=50


def test_embed_gates_log_softmax():
    from openviic_amd import ops
    g = torch.Generator().manual_seed(2)
    table, pos = torch.randn(53, 64, generator=g), torch.randn(7, 64, generator=g)
    tok, idx = torch.randint(0, 53, (5, 6), generator=g), torch.randint(0, 7, (5, 6), generator=g)
    got = ops.embed(tok.to(DEV), table.to(DEV), idx.to(DEV), pos.to(DEV))
    _close(got, table[tok] + pos[idx], tol=1e-7, what="embed")
    a, b2, c = (torch.randn(33, 64, generator=g) for _ in range(3))
    _close(ops.sigmoid_gate(a.to(DEV), b2.to(DEV)), a.double() * torch.sigmoid(b2.double()), what="sigmoid_gate")
    want = (c.double() + torch.sigmoid(a.double()) * b2.double()) / math.sqrt(3)
    _close(ops.gated_accumulate(c.to(DEV), a.to(DEV), b2.to(DEV), math.sqrt(3)), want, what="gated_accumulate")
    _close(ops.gated_accumulate(None, a.to(DEV), b2.to(DEV), 1.0), torch.sigmoid(a.double()) * b2.double(), what="gated first")
    x = torch.randn(17, 10201, generator=g) * 4
    _close(ops.log_softmax(x.to(DEV)), torch.log_softmax(x.double(), -1), tol=2e-6, what="log_softmax")


@pytest.mark.parametrize("trig", [False, True])
def test_box_relation_weights(trig):
    from openviic_amd import ops
    gold = golden("g5_box_relation.npz")
    boxes = torch.from_numpy(gold["boxes"])
    d_g, h = (16, 4) if trig else (4, 4)
    g = torch.Generator().manual_seed(12)
    w, b = torch.randn(h, d_g, generator=g), torch.randn(h, generator=g) * 0.1
    emb = torch.from_numpy(gold["trig" if trig else "plain"]).double()        # the reference's own embedding
    want = torch.relu(torch.einsum("bijd,hd->bhij", emb, w.double()) + b.double()[None, :, None, None])
    got = ops.box_relation_weights(boxes.to(DEV), w.to(DEV), b.to(DEV), trig)
    _close(got, want, tol=2e-5 if trig else 5e-6, what="box relation")
    _close(torch.from_numpy(gold["plain"]), box_relation_features(boxes, trignometric=False), tol=1e-6)


def _select_ref(logp, running, alive, k):
    B, W, V = logp.shape
    cand = running[:, :, None] + logp
    frozen = running[:, :, None].expand_as(cand).clone()
    frozen[:, :, 1:] = -999
    cand = alive[:, :, None] * cand + frozen * (1 - alive[:, :, None])
    val, idx = torch.sort(cand.view(B, -1), dim=-1, descending=True, stable=True)
    return idx[:, :k], val[:, :k], logp * alive[:, :, None]


@pytest.mark.parametrize("B,W,V,k", [(4, 1, 10201, 5), (7, 5, 10201, 5), (3, 3, 53, 3), (2, 8, 999, 8), (5, 1, 40, 1)])
def test_beam_select(B, W, V, k):
    from openviic_amd import native
    lib = native.load()
    g = torch.Generator().manual_seed(V + W)
    logp = torch.log_softmax(torch.randn(B, W, V, generator=g) * 3, -1)
    logp[0, 0, 5] = logp[0, 0, 9] = logp[0, 0].max() + 0.5                     # an exact tie at the top
    running = torch.randn(B, W, generator=g)
    alive = (torch.rand(B, W, generator=g) > 0.4).float()
    alive[:, 0] = 1
    want_idx, want_val, want_masked = _select_ref(logp, running, alive, k)
    d = lambda t: t.to(DEV).contiguous()
    lp, rn, al = d(logp), d(running), d(alive)
    chosen = torch.empty(B, k, dtype=torch.int64, device=DEV)
    score = torch.empty(B, k, device=DEV)
    masked = torch.empty_like(lp)
    scratch = torch.empty(8 * B * W * k, dtype=torch.uint8, device=DEV)
    args = (lp.data_ptr(), rn.data_ptr(), al.data_ptr(), B, W, V, k, chosen.data_ptr(), score.data_ptr())
    assert lib.ovc_beam_select(*args, masked.data_ptr(), scratch.data_ptr(), scratch.numel(), native.stream_handle()) == 0
    np.testing.assert_array_equal(chosen.cpu().numpy(), want_idx.numpy())
    np.testing.assert_array_equal(score.cpu().numpy(), want_val.numpy())       # same fp32 expression: bit-exact
    np.testing.assert_array_equal(masked.cpu().numpy(), want_masked.numpy())
    # without the masked log-probs a frozen beam's row is never read: same winners
    chosen.zero_(); score.zero_()
    assert lib.ovc_beam_select(*args, None, scratch.data_ptr(), scratch.numel(), native.stream_handle()) == 0
    np.testing.assert_array_equal(chosen.cpu().numpy(), want_idx.numpy())
    np.testing.assert_array_equal(score.cpu().numpy(), want_val.numpy())
    assert lib.ovc_beam_select(*args, None, scratch.data_ptr(), scratch.numel() - 8, native.stream_handle()) == -2      # OVC_EWORKSPACE


def test_beam_select_massive_ties_take_lowest_indices():
    """A live beam fed <pad> yields a uniform row: thousands of exactly equal candidates.  The engine's
    rule is "lower flat (beam, word) index first" (stable order); this drives the exhaustive fallback."""
    from openviic_amd import native
    lib = native.load()
    B, W, V, k = 3, 5, 10201, 5
    logp = torch.full((B, W, V), -9.230241775512695)          # -log(10201) in fp32, every word
    g = torch.Generator().manual_seed(4)
    logp[1] = torch.log_softmax(torch.randn(W, V, generator=g), -1)
    logp[1, 2] = -9.230241775512695                               # one uniform beam among normal ones
    running = torch.zeros(B, W)
    running[1] = torch.tensor([-3.0, -3.5, 9.0, -4.0, -2.0])       # the uniform beam leads: its words 0..4 win
    running[2] = torch.tensor([-1.0, -1.0, -1.0, -1.0, -1.0])
    alive = torch.ones(B, W)
    want_idx, want_val, _ = _select_ref(logp, running, alive, k)
    d = lambda t: t.to(DEV).contiguous()
    lp, rn, al = d(logp), d(running), d(alive)
    chosen = torch.empty(B, k, dtype=torch.int64, device=DEV)
    score = torch.empty(B, k, device=DEV)
    scratch = torch.empty(8 * B * W * k, dtype=torch.uint8, device=DEV)
    assert lib.ovc_beam_select(lp.data_ptr(), rn.data_ptr(), al.data_ptr(), B, W, V, k, chosen.data_ptr(),
                               score.data_ptr(), None, scratch.data_ptr(), scratch.numel(), native.stream_handle()) == 0
    np.testing.assert_array_equal(chosen.cpu().numpy(), want_idx.numpy())
    np.testing.assert_array_equal(score.cpu().numpy(), want_val.numpy())
    assert chosen[0].tolist() == [0, 1, 2, 3, 4] and chosen[1].tolist() == [2 * V + i for i in range(5)]


# ---- dual-collaborative (DLCT) embedding + encoder ---------------------------------------------------------

def _dlct_device_modules(emb_cfg, enc_cfg, emb_sd, enc_sd):
    from openviic_amd.builders import build_encoder, build_vision_embedding
    emb, enc = build_vision_embedding(emb_cfg).eval(), build_encoder(enc_cfg).eval()
    emb.load_state_dict(emb_sd)
    enc.load_state_dict(enc_sd)
    return emb.to(DEV), enc.to(DEV)


def _dlct_run(emb, enc, inputs):
    region, region_boxes, grid, grid_boxes = (t.to(DEV) for t in inputs)
    (rf, rm), (gf, gm), (r2a, g2a) = emb(region, region_boxes, grid, grid_boxes)
    out, mask = enc(rf, region_boxes, rm, r2a, gf, grid_boxes, gm, g2a)
    return rf, gf, rm, gm, r2a, g2a, out, mask


@pytest.mark.parametrize("trig", [False, True])
def test_dlct_encoder_against_reference_golden(trig):
    """G8: the reference's own sub-modules composed with the three documented mask repairs."""
    from helpers import dlct_case
    g = golden("g8_dlct_encoder%s.npz" % ("_trig" if trig else ""))
    emb_cfg, enc_cfg, emb_sd, enc_sd, inputs = dlct_case(trig)
    emb, enc = _dlct_device_modules(emb_cfg, enc_cfg, emb_sd, enc_sd)
    rf, gf, rm, gm, r2a, g2a, out, mask = _dlct_run(emb, enc, inputs)
    for got, name in ((rm, "region_mask"), (gm, "grid_mask"), (r2a, "region2all_mask"), (g2a, "grid2all_mask"),
                      (mask, "padding_mask")):
        np.testing.assert_array_equal(got.cpu().numpy(), g[name], err_msg=name)
    _close(rf, torch.from_numpy(g["region_embedded"]), what="region projection")
    _close(gf, torch.from_numpy(g["grid_embedded"]), what="grid projection")
    _close(out, torch.from_numpy(g["out"]), tol=5e-5, what="DLCT encoder output")
    n = g["region_mask"].shape[-1]
    assert (out[:, :n][rm[:, 0, 0]] == 0).all()          # padded region rows are cleared, as the repair says


@pytest.mark.parametrize("trig", [False, True])
def test_dlct_encoder_full_size_against_oracle(trig):
    """d=512, 50 regions + 7x7 grid cells (nk = 99), 3 layers, B=4, ragged regions."""
    from helpers import DLCT_FULL, dlct_case
    from oracle.dlct import OracleDualEncoder
    emb_cfg, enc_cfg, emb_sd, enc_sd, inputs = dlct_case(trig, DLCT_FULL, input_seed=17)
    orc = OracleDualEncoder(enc_cfg, emb_sd, enc_sd)
    region, region_boxes, grid, grid_boxes = inputs
    (orf, orm), (ogf, ogm), (or2a, og2a) = orc.embed(region, region_boxes, grid, grid_boxes)
    want, want_mask = orc.encode(orf, region_boxes, orm, or2a, ogf, grid_boxes, ogm, og2a)
    emb, enc = _dlct_device_modules(emb_cfg, enc_cfg, emb_sd, enc_sd)
    rf, gf, rm, gm, r2a, g2a, out, mask = _dlct_run(emb, enc, inputs)
    assert torch.equal(r2a.cpu(), or2a) and torch.equal(g2a.cpu(), og2a) and torch.equal(mask.cpu(), want_mask)
    assert orm.any() and not orm.all()
    _close(out, want, tol=1e-4, what="DLCT encoder (full size)")


def test_beam_select_random_shapes():
    """Selection against torch's stable sort on 40 seeded random shapes: widths 1..8, beams 1..8, vocabularies
    from 9 to 40003 words (every template instance of the row kernel and the streaming kernel), random frozen beams, planted exact ties."""
    from openviic_amd import native
    lib = native.load()
    rng = np.random.default_rng(2024)
    for case in range(40):
        W = int(rng.integers(1, 9))
        k = int(rng.integers(1, 9))
        V = int(rng.choice([9, 40, 257, 1000, 1024, 1025, 4096, 4100, 10201, 10240, 16384, 16385, 20000, 40003]))
        B = int(rng.integers(1, 6))
        if W * V < k:
            continue
        g = torch.Generator().manual_seed(1000 + case)
        logp = torch.log_softmax(torch.randn(B, W, V, generator=g) * float(rng.uniform(0.5, 4.0)), -1)
        running = torch.randn(B, W, generator=g) * 2
        alive = (torch.rand(B, W, generator=g) > 0.3).float()
        alive[:, 0] = 1
        if V > 12:                                   # exact ties inside a row and across rows
            logp[0, 0, 3] = logp[0, 0, 11] = logp[0, 0].max() + 0.25
            if W > 1:
                running[0, 1] = running[0, 0]
                logp[0, 1] = logp[0, 0]
                alive[0, 1] = 1
        want_idx, want_val, want_masked = _select_ref(logp, running, alive, k)
        d = lambda t: t.to(DEV).contiguous()
        lp, rn, al = d(logp), d(running), d(alive)
        chosen = torch.empty(B, k, dtype=torch.int64, device=DEV)
        score = torch.empty(B, k, device=DEV)
        masked = torch.empty_like(lp) if case % 2 else None
        scratch = torch.empty(8 * B * W * k, dtype=torch.uint8, device=DEV)
        rc = lib.ovc_beam_select(lp.data_ptr(), rn.data_ptr(), al.data_ptr(), B, W, V, k, chosen.data_ptr(), score.data_ptr(),
                                 None if masked is None else masked.data_ptr(), scratch.data_ptr(), scratch.numel(),
                                 native.stream_handle())
        assert rc == 0, (case, B, W, V, k)
        what = "case %d: B=%d W=%d V=%d k=%d" % (case, B, W, V, k)
        np.testing.assert_array_equal(chosen.cpu().numpy(), want_idx.numpy(), err_msg=what)
        np.testing.assert_array_equal(score.cpu().numpy(), want_val.numpy(), err_msg=what)
        if masked is not None:
            np.testing.assert_array_equal(masked.cpu().numpy(), want_masked.numpy(), err_msg=what)


def _tilings(lib):
    """[(index, kernel name, chains of its K-order class)] of every instance of the GEMM template."""
    import re
    out = []
    t = 0
    while lib.ovc_profile_kernel_name(t):
        name = lib.ovc_profile_kernel_name(t).decode()
        split = re.match(r"gemm_split_mfma<\d+, \d+, \d+, \d+, \d+, (\d+)>", name)
        if split:                                   # opt-in split-precision classes 103 / 104 (16-bit planes)
            out.append((t, name, 100 + int(split.group(1))))
        elif name.startswith("gemm_rows16_f32<"):   # the four-chain class on 16-row tiles (products of up to 112 rows)
            out.append((t, name, 4))
        else:
            wk, nc = (int(v) for v in re.match(r"gemm_f32_mfma<\d+, \d+, \d+, \d+, (\d+), \d+, (\d+)>", name).groups())
            out.append((t, name, wk * nc))
        t += 1
    return out


# max |err| / max |y| allowed against an fp64 product: fp32 MFMA classes, then three bf16 planes and two fp16 planes
# (gemm_split.h; the one- and two-plane bf16 classes 101 / 102 of round 2 failed the parity bar and were deleted)
CLASS_TOL = {1: 2e-5, 4: 2e-5, 103: 2e-5, 104: 2e-5}


def _linear_by_tiling(lib, native, xd, wd, bd, tiling, ksplit=1):
    M, K = xd.shape
    N = wd.shape[0]
    y = torch.empty((ksplit, M, N) if ksplit > 1 else (M, N), device=DEV)
    rc = lib.ovc_debug_linear_tiling(xd.data_ptr(), K, wd.data_ptr(), None if bd is None else bd.data_ptr(), y.data_ptr(),
                                     M, N, tiling, ksplit, 1, native.stream_handle())
    return rc, y


ROWS16_MAX_M = 112         # gemm.hip: kRows16MaxM


@pytest.mark.parametrize("M,N,K", [(130, 200, 96), (65, 33, 36), (1280, 512, 512), (257, 1536, 128), (31, 10201, 64),
                                   (640, 512, 2048), (5, 64, 32),
                                   # the decode products of B = 1 ... 12 at beam 5 (16-row instances, round 4), K tails and N tails
                                   (5, 512, 512), (40, 2048, 512), (16, 512, 2048), (33, 1536, 512), (7, 100, 36), (64, 48, 1028), (1, 16, 4), (112, 512, 512), (113, 512, 64), (100, 96, 64)])
def test_every_tiling_of_a_k_order_class_gives_the_same_bits(M, N, K):
    """The order in which a product sums over K is fixed per K-order class (gemm.hip): one chain, or four interleaved
    chains summed ((c0+c1)+c2)+c3 whether the chains live in one wave, two or four.  So every instance of the GEMM
    template must agree with fp64 within tolerance AND bit for bit with every other instance of its class -- which
    tiling a timing run picks can then never change a token id.  Shapes with M / N tails and (K = 36) a K tail."""
    from openviic_amd import native
    lib = native.load()
    lib.ovc_profile_kernel_name.restype = __import__("ctypes").c_char_p
    g = torch.Generator().manual_seed(M + N + K)
    x, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    want = x.double() @ w.double().T + b.double()
    xd, wd, bd = (t.to(DEV) for t in (x, w, b))
    tilings = _tilings(lib)
    assert len(tilings) == 33 and {c for _, _, c in tilings} == {1, 4, 103, 104}
    first = {}
    for t, name, chains in tilings:
        rc, got = _linear_by_tiling(lib, native, xd, wd, bd, t)
        if name.startswith("gemm_rows16") and M > ROWS16_MAX_M:
            assert rc != 0, name                      # the 16-row instances take products of up to 112 rows
            continue
        assert rc == 0, name
        _close(got, want, tol=CLASS_TOL[chains], what="%s on %dx%dx%d" % (name, M, N, K))
        if chains in first:
            assert torch.equal(got, first[chains][1]), "%s differs from %s (same K-order class)" % (name, first[chains][0])
        else:
            first[chains] = (name, got)
    if K >= 256:      # one chain and four chains are different summation orders: the classes are not interchangeable
        assert not torch.equal(first[1][1], first[4][1])
        err = {c: (first[c][1].cpu().double() - want).abs().max().item() for c in first}
        assert err[104] < 4 * err[1] and err[103] < 8 * err[1], err     # 22 / 24 operand bits: the fp32 path's own error level


def test_the_tile_map_is_exact_on_a_grid_of_65793_tiles():
    """The launcher hands the kernel its divisors as multiply-high magics (gemm.hip: TileMap, fast_div).  floor(2^32 / d) + 1 alone
    is exact only while tile * d < 2^32: on a grid of 3 x 21 931 = 65 793 tiles in one super-row group the LAST tile (65 792)
    would come out in group 1 -- outside the product, its corner of the output never written.  fast_div corrects the estimate
    once (it is never low and at most one high), so every element must be there.  (The smallest such grid the entry points'
    2 GB buffer limits admit; the model's own products have at most 12 800 tiles.)"""
    from openviic_amd import native
    lib = native.load()
    lib.ovc_profile_kernel_name.restype = __import__("ctypes").c_char_p
    K, N = 32, 32 * 21931 - 5
    d = 3 * 21931
    assert ((d - 1) * ((1 << 32) // d + 1)) >> 32 == 1           # the uncorrected estimate of (d - 1) / d
    wd = torch.randn(N, K, device=DEV) / math.sqrt(K)
    bd = torch.randn(N, device=DEV)
    ran = 0
    for t, name, chains in _tilings(lib):
        rows = {"gemm_f32_mfma<32, 32, 1, 1, 4, 32, 1>": 70, "gemm_f32_mfma<64, 32, 2, 1, 2, 32, 2>": 182}.get(name)
        if rows is None:
            continue                                  # the instances with 32-column tiles and a K tile of 32; three row tiles each
        xd = torch.randn(rows, K, device=DEV)
        y = torch.full((rows, N), float("nan"), device=DEV)
        rc = lib.ovc_debug_linear_tiling(xd.data_ptr(), K, wd.data_ptr(), bd.data_ptr(), y.data_ptr(), rows, N, t, 1, 1, native.stream_handle())
        assert rc == 0, name
        assert bool(torch.isfinite(y[-8:, -27:]).all()), "{}: the last tile was not written".format(name)
        worst = 0.0
        for c0 in range(0, N, 1 << 17):               # fp64 reference in column blocks (the whole product would be 1 GB)
            want = xd.double() @ wd[c0:c0 + (1 << 17)].double().T + bd[c0:c0 + (1 << 17)].double()
            worst = max(worst, float((y[:, c0:c0 + (1 << 17)].double() - want).abs().max()))
        assert worst < 1e-5, (name, worst)
        ran += 1
    assert ran == 2


@pytest.mark.parametrize("M,N,K,ksplit", [(130, 200, 96, 1), (65, 33, 48, 1), (1280, 512, 512, 2), (31, 10201, 64, 1), (640, 40, 2048, 4),
                                          (5, 64, 32, 1), (257, 1536, 128, 1)])
def test_pre_cut_weight_planes_give_the_bits_of_cutting_in_the_kernel(M, N, K, ksplit):
    """Split-precision classes: ovc_split_weight stores W as 16-bit planes in MFMA-operand order, and the GEMM instances that
    read them straight from memory must reproduce, bit for bit, the instances that cut W themselves through LDS -- for every
    tiling, with N tails (blocks past the last 32 rows of W), K not a multiple of the K tile (falls back) and K slices."""
    from openviic_amd import native
    lib = native.load()
    lib.ovc_profile_kernel_name.restype = __import__("ctypes").c_char_p
    g = torch.Generator().manual_seed(M * 3 + N + K)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    xd, wd, bd = (t.to(DEV) for t in (x, w, b))
    assert lib.ovc_split_weight_bytes(N, K + 4, 4) == 0 and lib.ovc_split_weight_bytes(N, K, 5) == 0      # K % 16, unknown mode
    assert lib.ovc_split_weight_bytes(N, K, 1) == 0 and lib.ovc_split_weight_bytes(N, K, 2) == 0          # modes deleted in round 3
    checked = 0
    for mode in (3, 4):
        planes = torch.empty(lib.ovc_split_weight_bytes(N, K, mode), dtype=torch.uint8, device=DEV)
        assert planes.numel() == ((N + 31) // 32) * (K // 16) * (2 if mode == 4 else mode) * 1024
        assert lib.ovc_split_weight(wd.data_ptr(), N, K, mode, planes.data_ptr(), native.stream_handle()) == 0
        for t, name, chains in _tilings(lib):
            if chains != 100 + mode:
                continue
            rc, want = _linear_by_tiling(lib, native, xd, wd, bd, t, ksplit)
            if rc != 0:           # a slice must be a whole number of this tiling's K tiles
                continue
            got = torch.empty_like(want)
            assert lib.ovc_debug_linear_planes(xd.data_ptr(), K, wd.data_ptr(), planes.data_ptr(), bd.data_ptr(), got.data_ptr(),
                                               M, N, t, ksplit, 1, native.stream_handle()) == 0, name
            assert torch.equal(got, want), name
            checked += 1
    assert checked >= 6


@pytest.mark.parametrize("M", [333, 40, 5])
@pytest.mark.parametrize("ksplit", [2, 4])
def test_k_slices_are_bit_identical_across_tilings_and_sum_to_the_product(ksplit, M):
    """K-split products (engine: the decode-step projections back to d_model): slice s is the product over
    k in [s K/ksplit, (s+1) K/ksplit); every tiling of the class writes the same partial products."""
    from openviic_amd import native
    lib = native.load()
    lib.ovc_profile_kernel_name.restype = __import__("ctypes").c_char_p
    N, K = 512, 1024
    g = torch.Generator().manual_seed(ksplit)
    x, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
    xd, wd = x.to(DEV), w.to(DEV)
    first = {}
    ran = set()
    for t, name, chains in _tilings(lib):
        rc, parts = _linear_by_tiling(lib, native, xd, wd, None, t, ksplit)
        if rc != 0:           # a slice must be a whole number of this tiling's K tiles; 16-row instances: up to 112 rows
            continue
        ran.add(name.split("<")[0])
        ks = K // ksplit
        for s_ in range(ksplit):
            _close(parts[s_], x[:, s_ * ks:(s_ + 1) * ks].double() @ w[:, s_ * ks:(s_ + 1) * ks].double().T, tol=CLASS_TOL[chains],
                   what="%s slice %d" % (name, s_))
        if chains in first:
            assert torch.equal(parts, first[chains])
        else:
            first[chains] = parts
    assert set(first) == {1, 4, 103, 104}
    assert ("gemm_rows16_f32" in ran) == (M <= ROWS16_MAX_M)


def test_tuner_only_ranks_inside_the_class_and_borrows_neighbouring_shapes():
    from openviic_amd import native
    lib = native.load()
    M, N, K = 192, 256, 128
    scratch = torch.randn(4 * (M * K + N * K + 4 * M * N) // 4 + 64, device=DEV)
    assert lib.ovc_debug_clear_tuning() == 0        # "near" is relative to MEASURED entries: none from earlier tests of this process
    for chains in (1, 4, 104):
        assert lib.ovc_gemm_tuned_get(M, N, 1, K, chains, 1, 1, 0) == -1
        calls = lib.ovc_gemm_tune_calls()
        assert lib.ovc_gemm_tune(M, N, 1, K, chains, 1, 1, 0, scratch.data_ptr(), scratch.numel() * 4, native.stream_handle()) == 0
        assert lib.ovc_gemm_tune_calls() == calls + 1
        t = lib.ovc_gemm_tuned_get(M, N, 1, K, chains, 1, 1, 0)
        assert t >= 0 and dict((i, c) for i, _, c in _tilings_cached(lib))[t] == chains
        assert lib.ovc_gemm_tune(M, N, 1, K, chains, 1, 1, 0, scratch.data_ptr(), scratch.numel() * 4, native.stream_handle()) == 0
        assert lib.ovc_gemm_tune_calls() == calls + 1                       # already measured: nothing runs
        assert lib.ovc_gemm_tuned_get(M + 40, N, 1, K, chains, 1, 1, 0) == -1     # exact look-up misses ...
        assert lib.ovc_gemm_tuned_get(M + 40, N, 1, K, chains, 1, 1, 1) == t      # ... the near one borrows the neighbour
        assert lib.ovc_gemm_tuned_get(4 * M, N, 1, K, chains, 1, 1, 1) == -1      # but not across more than a factor of two
        # single-segment products also borrow along the COLUMN count with M equal (the transposed vocabulary product: its batch
        # size is its seg_n) -- one axis at a time, within a factor of two
        assert lib.ovc_gemm_tuned_get(M, N + 40, 1, K, chains, 1, 1, 0) == -1 and lib.ovc_gemm_tuned_get(M, N + 40, 1, K, chains, 1, 1, 1) == t
        assert lib.ovc_gemm_tuned_get(M, N // 2, 1, K, chains, 1, 1, 1) == t and lib.ovc_gemm_tuned_get(M, 4 * N, 1, K, chains, 1, 1, 1) == -1
        assert lib.ovc_gemm_tuned_get(M + 40, N + 40, 1, K, chains, 1, 1, 1) == -1
        assert lib.ovc_gemm_tuned_get(M, 64, 3, K, chains, 1, 1, 1) == -1           # segmented products: M only
        # every objective has its own table, named in the call (ABI 6: no process-wide "current objective")
        assert lib.ovc_gemm_tuned_get(M, N, 1, K, chains, 1, 2, 1) == -1
        assert lib.ovc_gemm_tuned_set(M, N, 1, K, chains, 1, 2, t) == 0 and lib.ovc_gemm_tuned_get(M, N, 1, K, chains, 1, 2, 0) == t
        wrong = next(i for i, _, c in _tilings_cached(lib) if c != chains)
        assert lib.ovc_gemm_tuned_set(M, N, 1, K, chains, 1, 1, wrong) != 0    # a tiling of the other class is refused
        assert lib.ovc_gemm_tuned_set(M, N, 1, K, chains, 1, 9, t) != 0        # objectives are 1..8
    assert lib.ovc_gemm_tune(M, N, 1, K, 102, 1, 1, 0, scratch.data_ptr(), scratch.numel() * 4, native.stream_handle()) != 0   # deleted class
    assert lib.ovc_gemm_tuned_set(16, 512, 3, 2048, 4, 2, 1, 7) != 0           # segmented outputs cannot split
    assert lib.ovc_gemm_tuned_set(16, 512, 1, 2048, 4, 8, 1, 7) != 0           # more than 4 slices


def _tilings_cached(lib):
    lib.ovc_profile_kernel_name.restype = __import__("ctypes").c_char_p
    return _tilings(lib)


def test_beam_select_with_nan_scores_returns_in_range_indices():
    """NaN log-probs (an image whose every attention key is masked) have no order; the indices handed back must
    still be usable as gather indices by the step-wise host loop."""
    from openviic_amd import native
    lib = native.load()
    B, W, V, k = 2, 3, 200, 3
    logp = torch.log_softmax(torch.randn(B, W, V, generator=torch.Generator().manual_seed(3)), -1)
    logp[1] = float("nan")
    running, alive = torch.zeros(B, W), torch.ones(B, W)
    want_idx, want_val, _ = _select_ref(logp[:1], running[:1], alive[:1], k)
    d = lambda t: t.to(DEV).contiguous()
    lp, rn, al = d(logp), d(running), d(alive)
    chosen = torch.empty(B, k, dtype=torch.int64, device=DEV)
    score = torch.empty(B, k, device=DEV)
    scratch = torch.empty(8 * B * W * k, dtype=torch.uint8, device=DEV)
    assert lib.ovc_beam_select(lp.data_ptr(), rn.data_ptr(), al.data_ptr(), B, W, V, k, chosen.data_ptr(), score.data_ptr(),
                               None, scratch.data_ptr(), scratch.numel(), native.stream_handle()) == 0
    np.testing.assert_array_equal(chosen[:1].cpu().numpy(), want_idx.numpy())
    assert chosen[1].min() >= 0 and chosen[1].max() < W * V


def test_embed_clamps_indices_instead_of_reading_out_of_bounds():
    from openviic_amd import ops
    table = torch.randn(10, 8, generator=torch.Generator().manual_seed(1))
    pos = torch.randn(4, 8, generator=torch.Generator().manual_seed(2))
    tokens = torch.tensor([[0, 9, 10, 12345678901, -3]])
    positions = torch.tensor([[0, 3, 4, 99, -1]])
    got = ops.embed(tokens.to(DEV), table.to(DEV), positions.to(DEV), pos.to(DEV)).cpu()
    want = table[tokens.clamp(0, 9)] + pos[positions.clamp(0, 3)]
    np.testing.assert_array_equal(got.numpy(), want.numpy())


def test_operator_wrappers_reject_mismatched_shapes():
    """The kernels index with the caller's sizes; a tensor of another shape must raise on the host, not read
    device memory out of bounds."""
    from openviic_amd import native, ops
    t = lambda *shape: torch.randn(*shape, device=DEV)
    with pytest.raises(native.OvcError):
        ops.linear(t(6, 32), t(16, 32), t(15))                               # bias length
    with pytest.raises(native.OvcError):
        ops.linear(t(6, 32), t(16, 32), residual=t(5, 16))                   # residual rows
    with pytest.raises(native.OvcError):
        ops.linear(t(6, 32), t(16, 40), x2=t(5, 8))                          # second input block rows
    with pytest.raises(native.OvcError):
        ops.layer_norm(t(4, 32), t(31), t(32))                               # gamma length
    with pytest.raises(native.OvcError):
        ops.layer_norm(t(4, 32), t(32), t(32), residual=t(3, 32))            # residual shape
    with pytest.raises(native.OvcError):
        ops.attention(t(2, 5, 32), t(2, 7, 32), t(2, 6, 32), 4)              # k / v disagree on the key count
    with pytest.raises(native.OvcError):
        ops.attention(t(2, 5, 32), t(2, 7, 32), t(2, 7, 32), 4, geometry=t(2, 4, 5, 6))
    with pytest.raises(native.OvcError):
        ops.sigmoid_gate(t(8, 16), t(8, 12))
    with pytest.raises(native.OvcError):
        ops.box_relation_weights(t(2, 5, 3), t(4, 4), t(4), False)           # boxes need 4 coordinates
    with pytest.raises(native.OvcError):
        ops.beam_select(t(2, 3, 50), t(2, 3), torch.ones(2, 3, 1, device=DEV), None, 2, 9)   # beam wider than the ABI's maximum
    assert ops.linear(t(6, 32), t(16, 32), t(16)).shape == (6, 16)          # the well-formed call still works


def _fused_select(lib, native, x, fc, running, alive, B, W, V, k, transposed, kchains=None, tiling=-1):
    """One selection step through the fused path.  ``kchains``: the product's fp32 K-order class (default: what the engine
    runs -- one chain for the transposed form, four for the row-major one); ``tiling`` pins one instance of that class."""
    chosen = torch.empty(B, k, dtype=torch.int64, device=DEV)
    score = torch.empty(B, k, device=DEV)
    scratch = torch.empty(lib.ovc_debug_vocab_select_bytes(B, W, V, k), dtype=torch.uint8, device=DEV)
    kchains = kchains or (1 if transposed else 4)
    assert lib.ovc_debug_force_gemm_tiling(tiling) == 0
    try:
        rc = lib.ovc_debug_vocab_select(x.data_ptr(), fc.data_ptr(), running.data_ptr(), alive.data_ptr(), B, W, V, x.shape[1], k,
                                        int(transposed), kchains, scratch.data_ptr(), scratch.numel(), chosen.data_ptr(), score.data_ptr(),
                                        native.stream_handle())
    finally:
        lib.ovc_debug_force_gemm_tiling(-1)
    assert rc == 0
    return chosen.cpu(), score.cpu()


def _tilings_of_class(lib, chains):
    out, t = [], 0
    while lib.ovc_profile_kernel_name(t):
        name = lib.ovc_profile_kernel_name(t).decode()
        if name.startswith("gemm_f32_mfma<"):
            bm, bn, wm, wn, wk, bk, nc = (int(v) for v in name[len("gemm_f32_mfma<"):-1].split(","))
            if wk * nc == chains:
                out.append(t)
        t += 1
    return out


@pytest.mark.parametrize("B,W,V,k,d", [(6, 5, 10201, 5, 64), (9, 1, 10201, 5, 64), (3, 3, 53, 3, 32), (4, 8, 999, 8, 64),
                                       (5, 5, 16384, 5, 32), (2, 2, 40, 2, 32)])
def test_fused_selection_from_block_pieces_matches_a_stable_sort(B, W, V, k, d):
    """The engine's selection never reads the logits back: the vocabulary GEMM's epilogue leaves (max, sum exp) per 32-word
    block and beam_fused_update_kernel picks the image's k winners from those pieces plus a few gathered blocks
    (reference semantics: models/modules/beam_search.py:45-59).  Operands with small dyadic values make every logit EXACT in
    fp32, so the winners must be those of a stable descending sort -- ties inside a row broken by the lower word index --
    for live and frozen beams, in both orientations of the product (transposed = the fp32 engine's, row-major = the
    split-precision modes')."""
    from openviic_amd import native
    lib = native.load()
    g = torch.Generator().manual_seed(V + 7 * W + B)
    x = torch.randint(-4, 5, (B * W, d), generator=g).float() / 8
    fc = torch.randint(-2, 3, (V, d), generator=g).float() / 2
    fc[min(40, V - 1)] = fc[7]                     # exact ties inside every row: the lower word index must win
    fc[V - 1] = fc[3]
    running = torch.randn(B, W, generator=g)
    alive = (torch.rand(B, W, generator=g) > 0.3).float()
    alive[:, 0] = 1
    logits = x.double() @ fc.double().T
    assert torch.equal(logits.float().double(), logits)                       # exact in fp32
    logp = torch.log_softmax(logits, -1).view(B, W, V)
    cand = running.double()[:, :, None] + logp
    frozen = torch.full_like(cand, -999.0)
    frozen[:, :, 0] = running.double()
    cand = torch.where(alive[:, :, None] > 0, cand, frozen)
    want_val, want_idx = torch.sort(cand.view(B, -1), dim=-1, descending=True, stable=True)
    margin = (want_val[:, :k] - want_val[:, 1:k + 1]).abs()
    got = {}
    # (orientation, K-order class): what the fp32 engine runs -- transposed, ONE chain (the hook used to run four) --
    # then the four-chain transposed form and the row-major form of the split-precision modes
    for transposed, chains in ((1, 1), (1, 4), (0, 4)):
        idx, val = _fused_select(lib, native, x.to(DEV), fc.to(DEV), running.to(DEV).contiguous(), alive.to(DEV).contiguous(), B, W, V, k,
                                 transposed, chains)
        got[(transposed, chains)] = (idx, val)
        np.testing.assert_allclose(val.numpy(), want_val[:, :k].numpy(), rtol=0, atol=2e-5)
        # winners: identical wherever the next candidate is not within fp32 noise of the chosen one, or ties exactly with it
        # inside one row (then the order of the stable sort is the rule)
        for b in range(B):
            for j in range(k):
                same_row_tie = j + 1 < W * V and want_idx[b, j] // V == want_idx[b, j + 1] // V and margin[b, j] == 0
                prev_tie = j > 0 and want_idx[b, j] // V == want_idx[b, j - 1] // V and margin[b, j - 1] == 0
                if margin[b, j] > 1e-5 and (j == 0 or margin[b, j - 1] > 1e-5) or same_row_tie or prev_tie:
                    if (j == 0 or margin[b, j - 1] > 1e-5 or prev_tie) and (margin[b, j] > 1e-5 or same_row_tie):
                        assert idx[b, j] == want_idx[b, j], (b, j, idx[b], want_idx[b, :k + 1], want_val[b, :k + 1])
    # the orientations / classes sum a block's exponentials in different (each fixed) orders: same winners, scores to rounding
    for other in ((1, 4), (0, 4)):
        assert torch.equal(got[(1, 1)][0], got[other][0])
        np.testing.assert_allclose(got[(1, 1)][1].numpy(), got[other][1].numpy(), rtol=0, atol=2e-6)
    # every tiling of the production class (one chain: 128x128 ... 64x64, K tiles 16 / 32 / 64) leaves the same pieces:
    # winners AND scores bit-identical whichever instance's stats_t epilogue ran
    for tiling in _tilings_of_class(lib, 1):
        idx, val = _fused_select(lib, native, x.to(DEV), fc.to(DEV), running.to(DEV).contiguous(), alive.to(DEV).contiguous(), B, W, V, k,
                                 1, 1, tiling)
        assert torch.equal(idx, got[(1, 1)][0]) and torch.equal(val, got[(1, 1)][1]), tiling


def test_fused_selection_massive_ties_frozen_beams_and_nan_rows():
    """Uniform rows (a live beam fed <pad>: thousands of exactly equal candidates -> the exhaustive fallback), images whose
    beams are all frozen, and NaN rows (an image without valid regions) -- lowest flat index first, in-range indices always."""
    from openviic_amd import native
    lib = native.load()
    B, W, V, k, d = 4, 5, 10201, 5, 32
    g = torch.Generator().manual_seed(3)
    x = torch.randint(-4, 5, (B * W, d), generator=g).float() / 8
    fc = torch.randint(-2, 3, (V, d), generator=g).float() / 2
    x[0:W] = 0                                          # image 0: every row uniform -> all candidates of a row tie
    running = torch.tensor([[0.0, -1.0, -2.0, -3.0, -4.0]] * B)
    alive = torch.ones(B, W)
    alive[1] = 0                                        # image 1: every beam frozen
    x[2 * W:3 * W] = float("nan")                       # image 2: NaN logits
    idx, val = _fused_select(lib, native, x.to(DEV), fc.to(DEV), running.to(DEV), alive.to(DEV), B, W, V, k, 1)
    np.testing.assert_array_equal(idx[0].numpy(), np.arange(k))                         # beam 0 (best running), words 0..k-1
    np.testing.assert_allclose(val[0].numpy(), -np.log(V), rtol=1e-6)
    # frozen beams offer word 0 at their running score and -999 for every other word: the k beams' word 0, best running first
    np.testing.assert_array_equal(idx[1].numpy(), np.arange(k) * V)
    np.testing.assert_array_equal(val[1].numpy(), running[1].numpy())
    assert ((idx[2] >= 0) & (idx[2] < W * V)).all()                                       # arbitrary but in range
    logits = x[3 * W:].double() @ fc.double().T
    cand = (running[3].double()[:, None] + torch.log_softmax(logits, -1)).view(-1)
    want_val, want_idx = torch.sort(cand, descending=True, stable=True)
    np.testing.assert_allclose(val[3].numpy(), want_val[:k].numpy(), rtol=0, atol=2e-5)
