"""Seeded random sweep of the fused engine against the CPU oracle (test infrastructure: ``oracle/``).

The fixed parity cases sit at the BASELINE sizes and at hand-picked limits (``test_engine_gpu.py``); this sweep walks the space
in between: every architecture, head counts / head sizes / widths / layer counts / memory slots / vocabulary sizes / beam widths /
ragged region counts drawn at random (sizes the oracle finishes in well under a second), each decoded by the engine and by the
oracle's restatement of the reference op sequence.  Bar as everywhere: padding mask bit-exact, encoder output within 2e-4 / 2e-5
(the object-relation encoder, whose geometry bias is a log next to a ReLU's zero: relative L2 error of the output),
token ids identical for every image whose decision margins in the oracle run exceed fp32 noise, log-probabilities within 1e-3.

``OVC_FUZZ_CASES=n`` runs n cases (default 40, ~20 s); ``OVC_FUZZ_SEED`` moves the stream.
"""
import os
import random

import numpy as np
import pytest
import torch

from helpers import VARIANTS, batch, decided_images, device_model, teacher_tokens
from openviic_amd import native
from oracle.captioner import OracleCaptioner

pytestmark = pytest.mark.gpu
MARGIN = 2e-4

# (heads, d_k) with heads * d_k a multiple of the 64-wide tile (include/ovc.h, model_ok) and d_k a power of two in 4..64
HEAD_SHAPES = [(1, 64), (2, 32), (2, 64), (3, 64), (4, 16), (4, 32), (4, 64), (6, 32), (8, 8), (8, 16), (8, 32), (16, 4), (16, 8), (12, 16)]


def _draw(rng):
    variant = rng.choice(VARIANTS)
    heads, d_kv = rng.choice(HEAD_SHAPES)
    dims = dict(d_feature=4 * rng.randint(2, 40), d_model=32 * rng.randint(1, 8) if rng.random() < 0.7 else 4 * rng.randint(5, 60),
                heads=heads, d_kv=d_kv, d_ff=4 * rng.randint(4, 96), layers=rng.randint(1, 4))
    if variant in ("attention_on_attention", "meshed_memory_transformer") and rng.random() < 0.85:
        dims["d_model"] = 32 * rng.randint(1, 8)                         # two-block products: d_model a multiple of 32 (else refused)
    if variant == "meshed_memory_transformer":
        dims["memory"] = rng.choice([1, 3, 8, 17, 40])
    # region counts on both sides of the edge (128 regions / 192 keys with the memory slots) between the register-resident and the key-tiled
    # attention instances; round 3's draw stopped at 65 and could not see that 89..128 regions + 40 slots failed late
    B, N = rng.randint(1, 7), rng.choice([1, 2, 3, 5, 8, 13, 16, 17, 31, 32, 33, 50, 64, 65, 88, 89, 100, 127, 128, 129, 150, 196, 257])
    V = rng.choice([5, 9, 33, 64, 100, 257, 1000, 4099])
    T = rng.randint(2, 9)
    k = rng.randint(1, min(5, V - 1))
    # encoders.py:93-101: the trigonometric embedding has d_model / heads features, 8 per frequency (sin, cos of 4 box relations)
    trig = variant == "object_relation_transformer" and rng.random() < 0.5 and (dims["d_model"] // heads) % 8 == 0 and dims["d_model"] % heads == 0
    return variant, dims, (B, N, V, T, k), trig


def test_random_architectures_and_shapes_against_the_oracle():
    from openviic_amd.builders import build_model
    from openviic_amd.config import model_config
    from openviic_amd.utils.synthetic import SyntheticVocab, synthetic_boxes, synthetic_features, synthetic_state_dict
    cases = int(os.environ.get("OVC_FUZZ_CASES", "40"))
    rng = random.Random(int(os.environ.get("OVC_FUZZ_SEED", "20261004")))
    checked_images = decided_total = refused = 0
    worst_conditioning = 0.0
    for case in range(cases):
        variant, dims, (B, N, V, T, k), trig = _draw(rng)
        what = "case {}: {} {} B={} N={} V={} T={} k={} trig={}".format(case, variant, dims, B, N, V, T, k, trig)
        vocab = SyntheticVocab(V, T)
        cfg = model_config(variant, device="cpu", trignometric_embedding=trig, **dims)
        sd = synthetic_state_dict(build_model(cfg, vocab).state_dict(), seed=1000 + case, mode="generic",
                                  memory_dims=(dims["d_kv"], dims.get("memory", 40)))
        feats = synthetic_features(B, N, dims["d_feature"], seed=case, ragged=True)
        boxes = synthetic_boxes(B, N, seed=case) if variant == "object_relation_transformer" else None
        orc = OracleCaptioner(cfg, sd, V, T)
        rec = {}
        want_ids, want_logp = orc.beam_search(feats, k, out_size=k, boxes=boxes, record=rec)
        want_enc, want_mask = orc.encode(feats, boxes)
        model = device_model(cfg, vocab, sd)
        if variant in ("attention_on_attention", "meshed_memory_transformer") and dims["d_model"] % 32:
            # documented limit (include/ovc.h): two-block products need the seam on a K-tile boundary -- refused up front
            with pytest.raises(native.OvcError, match="unsupported"):
                model.beam_search(batch(feats, boxes), batch_size=B, beam_size=k, out_size=k)
            refused += 1
            continue
        try:
            with torch.no_grad():
                ids, logp = model.beam_search(batch(feats, boxes), batch_size=B, beam_size=k, out_size=k)
                enc, mask = model.encoder_forward(batch(feats, boxes))
        except Exception as error:                                      # name the case: the engine's error codes carry no shapes
            raise AssertionError("{}: {}".format(what, error)) from error
        assert torch.equal(mask.cpu(), want_mask), what
        live = ~want_mask.reshape(B, -1).all(dim=1).numpy()            # an image without any region decodes to NaN in the reference
        got_enc, ref_enc = enc.cpu().numpy()[live], want_enc.numpy()[live]
        if variant == "object_relation_transformer":
            # The geometry bias is log(clamp(relu(fc_g(box relations)), 1e-6)) (attentions.py:97-114): next to the ReLU's zero a
            # rounding-level change of fc_g's output moves the bias by O(1), and the trigonometric embedding takes sin / cos of
            # angles up to ~700 rad (one fp32 ulp of the angle: 6e-5).  This encoder is ill-conditioned, so it is held to a
            # CONDITIONING statement instead of a constant (the constant had grown 3e-4 -> 3e-3 on an
            # argument): the same operation sequence in fp64 on the same fp32 weights and inputs is the yardstick, and the HIP
            # result may be no further from it than three times what the reference's own fp32 arithmetic (the oracle) is --
            #     ||HIP - fp64|| <= max(floor, 3 ||oracle_fp32 - fp64||)        (relative L2 norms of the encoder output)
            # with the floor at the constants this test held before round 3 (3e-4 trigonometric, 5e-5 plain): below them one
            # case's ratio is one draw of "which elements flipped across the ReLU's zero" in either implementation and says
            # nothing (sweep seed 43, case 81: HIP 1.9e-4, oracle 5.0e-5, both tiny).  Measured on 400 + 150 random
            # models (tools/trig_conditioning_probe.py, profiles/r04_trig_conditioning.txt): the two fp32 results sit equally far
            # from fp64 -- medians 1.72e-5 / 1.62e-5, maxima 9.2e-4 / 1.07e-3 with the trigonometric embedding, 4e-7 / 1e-5
            # without; wherever the HIP error exceeds 1e-4 the ratio is 0.85 .. 1.39.  Element-wise only a coarse sanity bound
            # (a handful of elements per thousand sit next to the ReLU's zero).
            ref64 = OracleCaptioner(cfg, sd, V, T, dtype=torch.float64).encode(feats, boxes)[0].numpy()[live]
            scale = max(np.linalg.norm(ref64), 1e-300)
            e_hip, e_cpu = np.linalg.norm(got_enc - ref64) / scale, np.linalg.norm(ref_enc - ref64) / scale
            floor = 3e-4 if trig else 5e-5
            assert e_hip <= max(floor, 3.0 * e_cpu), "{}: HIP {:.2e} from the fp64 result, the fp32 oracle {:.2e}".format(what, e_hip, e_cpu)
            worst_conditioning = max(worst_conditioning, e_cpu)
            np.testing.assert_allclose(got_enc, ref_enc, rtol=5e-2, atol=2e-2, err_msg=what)
        elif N > 128:
            # Key-tiled attention (round 4).  With hundreds of keys and a narrow model the reference's OWN fp32 arithmetic is no
            # longer within 2e-5 of the exact result element-wise (d_model 64, d_ff 16, 3 layers, 257 regions: the fp32 oracle
            # is 1.3e-5 from its fp64 self at the worst element, so two correct fp32 implementations may differ by twice that;
            # sweep seed 44 case 51 had ONE element of 296 064 at 3.3e-5).  So the statement is made against fp64 here too,
            # for the worst element and for the norm: HIP at most three times as far from it as the fp32 oracle is.
            ref64 = OracleCaptioner(cfg, sd, V, T, dtype=torch.float64).encode(feats, boxes)[0].numpy()[live]
            d_hip, d_cpu = np.abs(got_enc - ref64).max(), np.abs(ref_enc - ref64).max()
            assert d_hip <= max(2e-5, 3.0 * d_cpu), "{}: worst element HIP {:.2e} from fp64, the fp32 oracle {:.2e}".format(what, d_hip, d_cpu)
            scale = max(np.linalg.norm(ref64), 1e-300)
            e_hip, e_cpu = np.linalg.norm(got_enc - ref64) / scale, np.linalg.norm(ref_enc - ref64) / scale
            assert e_hip <= max(2e-6, 3.0 * e_cpu), "{}: HIP {:.2e} from the fp64 result, the fp32 oracle {:.2e}".format(what, e_hip, e_cpu)
            np.testing.assert_allclose(got_enc, ref_enc, rtol=2e-4, atol=2e-5 + 2.0 * d_cpu, err_msg=what)
        else:
            np.testing.assert_allclose(got_enc, ref_enc, rtol=2e-4, atol=2e-5, err_msg=what)
        assert int(ids.min()) >= 0 and int(ids.max()) < V, what
        gaps, inner = torch.stack(rec["gap"]).numpy(), torch.stack(rec["inner_gap"]).numpy()
        decided = decided_images(gaps, inner, MARGIN) & live
        if k > 1:
            decided &= inner[-1].min(axis=1) > MARGIN                   # out_size = k: the whole final order counts
        checked_images += B
        decided_total += int(decided.sum())
        np.testing.assert_array_equal(ids.cpu().numpy()[decided], want_ids.numpy()[decided], err_msg=what)
        got, want = logp.cpu().numpy()[decided], want_logp.numpy()[decided]
        finite = np.isfinite(want)
        assert np.array_equal(np.isfinite(got), finite), what
        np.testing.assert_allclose(got[finite], want[finite], rtol=0, atol=1e-3, err_msg=what)
        # every third case also takes the other entry points: the teacher-forced forward (operator path) against the oracle's,
        # return_probs (every word's masked log-probability at every step) and out_size = 1 against the calls above
        if case % 3 == 0:
            tokens = teacher_tokens(B, T, V, seed=case, with_pad=T >= 4) if V > 4 else torch.ones(B, T, dtype=torch.long)
            want_fwd = orc.forward(feats, tokens, boxes)
            with torch.no_grad():
                fwd = model(batch(feats, boxes, tokens))
                ids_p, logp_p, everything = model.beam_search(batch(feats, boxes), batch_size=B, beam_size=k, out_size=k, return_probs=True)
                ids_1, logp_1 = model.beam_search(batch(feats, boxes), batch_size=B, beam_size=k, out_size=1)
            np.testing.assert_allclose(fwd.cpu().numpy()[live], want_fwd.numpy()[live], rtol=0, atol=1e-3, err_msg=what + " (forward)")
            assert torch.equal(ids_p, ids) and torch.equal(logp_p, logp), what + " (return_probs changes the captions)"
            assert tuple(everything.shape) == (B, k, T, V), what
            ids_k, logp_k = ids.reshape(B, k, T), logp.reshape(B, k, T)            # out_size = 1 drops the beam axis
            want_all = orc.beam_search(feats, k, out_size=k, return_probs=True, boxes=boxes)[2]
            np.testing.assert_allclose(everything.cpu().numpy()[decided], want_all.numpy()[decided], rtol=0, atol=1e-3,
                                       err_msg=what + " (return_probs)")
            assert torch.equal(ids_1, ids_k[:, 0]) and torch.equal(logp_1, logp_k[:, 0]), what + " (out_size = 1)"
        model._engine.release()
    print("[fuzz] {} cases ({} refused up front), {} images, {} decided ({:.0f} %); worst-conditioned object-relation encoder: "
          "fp32 oracle {:.1e} from fp64".format(cases, refused, checked_images, decided_total,
                                                100.0 * decided_total / max(1, checked_images), worst_conditioning))
    assert decided_total >= 0.5 * checked_images


def test_random_call_sequences_replay_graphs_like_plain_launches():
    """A random walk over shapes, streams and output options on ONE engine with hipGraph replay on (more distinct shapes than the
    graph cache holds, so entries are evicted and re-captured; two streams with their own workspaces that grow along the way; the
    same shape revisited with other features) against an engine that only ever launches plainly: every call bit-identical."""
    from openviic_amd.engine import CaptionEngine
    from helpers import TINY, tiny_case
    from openviic_amd.utils.synthetic import synthetic_features
    lib = native.load()
    rng = random.Random(int(os.environ.get("OVC_FUZZ_SEED", "20261004")) + 7)
    cfg, vocab, sd, _, _ = tiny_case("meshed_memory_transformer")
    model = device_model(cfg, vocab, sd)
    graphed, plain = CaptionEngine(model), CaptionEngine(model)
    graphed.use_graph, plain.use_graph = True, False
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    shapes = [(B, N, k) for B in (1, 2, 3, 5) for N in (1, 4, 7, 9, 16) for k in (1, 2, 3)]          # 60 shapes > OVC_GRAPH_CACHE_MAX
    calls = int(os.environ.get("OVC_FUZZ_CASES", "40")) * 6
    peak = 0
    for call in range(calls):
        B, N, k = rng.choice(shapes[:12]) if rng.random() < 0.5 else rng.choice(shapes)             # a hot set plus a long tail
        feats = synthetic_features(B, N, TINY["d_feature"], seed=rng.randint(0, 5), ragged=True).cuda()
        out_size = rng.choice([1, k])
        probs = rng.random() < 0.15
        stream = rng.choice(streams)
        with torch.no_grad(), torch.cuda.stream(stream):
            got = graphed.beam_search(feats, None, B, k, out_size=out_size, return_probs=probs)
            want = plain.beam_search(feats, None, B, k, out_size=out_size, return_probs=probs)
        stream.synchronize()
        for g_t, w_t in zip(got, want):
            assert torch.equal(g_t, w_t), "call {}: B={} N={} k={} out_size={} return_probs={}".format(call, B, N, k, out_size, probs)
        peak = max(peak, lib.ovc_graph_cache_size())
    assert 0 < peak <= int(os.environ.get("OVC_GRAPH_CACHE_MAX", "24"))
    graphed.release(); plain.release()


def test_random_dual_collaborative_encoders_against_the_oracle():
    """the dual-collaborative (DLCT) embedding + encoder on random sizes -- region counts, grid sizes
    (n + g*g keys up to 244: both sides of the register / key-tiled edge of the attention instances), feature widths, head shapes, layers, both box embeddings -- against ``oracle/dlct.py``: the five
    masks bit-exact, the encoder output by its relative L2 error (its geometry bias is the object-relation one, see above)."""
    from helpers import dlct_case
    from openviic_amd.builders import build_encoder, build_vision_embedding
    from oracle.dlct import OracleDualEncoder
    rng = random.Random(int(os.environ.get("OVC_FUZZ_SEED", "20261004")) + 13)
    cases = max(4, int(os.environ.get("OVC_FUZZ_CASES", "40")) // 4)
    for case in range(cases):
        heads, d_kv = rng.choice([(2, 32), (4, 16), (4, 32), (8, 8), (8, 16), (2, 64), (1, 64)])
        trig = rng.random() < 0.5
        d_model = heads * 8 * rng.randint(1, 4) if trig else 32 * rng.randint(1, 6)      # trigonometric: d_model / heads a multiple of 8
        if d_model % 32:
            d_model = 32 * (d_model // 32 + 1) if not trig else d_model
        grid = rng.randint(1, 12)                                    # regions + cells: up to 100 + 144 keys (key-tiled beyond 128)
        shape = dict(B=rng.randint(1, 4), n_regions=rng.randint(2, 100 if grid > 7 else min(60, 128 - grid * grid)), grid=grid,
                     d_region=4 * rng.randint(2, 30), d_grid=4 * rng.randint(2, 30), d_model=d_model, heads=heads, d_kv=d_kv,
                     d_ff=4 * rng.randint(4, 64), layers=rng.randint(1, 3))
        what = "case {}: trig={} {}".format(case, trig, shape)
        if trig and (d_model % heads or (d_model // heads) % 8):
            continue
        emb_cfg, enc_cfg, emb_sd, enc_sd, inputs = dlct_case(trig, shape, input_seed=100 + case)
        orc = OracleDualEncoder(enc_cfg, emb_sd, enc_sd)
        region, region_boxes, grid_f, grid_boxes = inputs
        (orf, orm), (ogf, ogm), (or2a, og2a) = orc.embed(region, region_boxes, grid_f, grid_boxes)
        want, want_mask = orc.encode(orf, region_boxes, orm, or2a, ogf, grid_boxes, ogm, og2a)
        emb, enc = build_vision_embedding(emb_cfg).eval(), build_encoder(enc_cfg).eval()
        emb.load_state_dict(emb_sd)
        enc.load_state_dict(enc_sd)
        emb, enc = emb.to("cuda"), enc.to("cuda")
        try:
            with torch.no_grad():
                (rf, rm), (gf, gm), (r2a, g2a) = emb(*(t.to("cuda") for t in inputs))
                out, mask = enc(rf, region_boxes.to("cuda"), rm, r2a, gf, grid_boxes.to("cuda"), gm, g2a)
        except Exception as error:
            raise AssertionError("{}: {}".format(what, error)) from error
        assert torch.equal(rm.cpu(), orm) and torch.equal(gm.cpu(), ogm), what
        assert torch.equal(r2a.cpu(), or2a) and torch.equal(g2a.cpu(), og2a) and torch.equal(mask.cpu(), want_mask), what
        got, ref = out.cpu().numpy(), want.numpy()
        keep = np.isfinite(ref).all(axis=-1)                        # a row whose every key is masked is NaN in the reference
        # the same conditioning statement as for the object-relation encoder above: fp64 on the same weights and inputs is the
        # yardstick, the HIP result may be at most three times as far from it as the fp32 oracle is (floors 3e-4 / 5e-5, the
        # constants of the rounds before the bound was loosened; profiles/r04_trig_conditioning.txt has this encoder's numbers)
        orc64 = OracleDualEncoder(enc_cfg, emb_sd, enc_sd, dtype=torch.float64)
        (orf64, _), (ogf64, _), _ = orc64.embed(region, region_boxes, grid_f, grid_boxes)
        ref64 = orc64.encode(orf64, region_boxes, orm, or2a, ogf64, grid_boxes, ogm, og2a)[0].numpy()
        scale = max(np.linalg.norm(ref64[keep]), 1e-300)
        e_hip, e_cpu = np.linalg.norm(got[keep] - ref64[keep]) / scale, np.linalg.norm(ref[keep] - ref64[keep]) / scale
        assert e_hip <= max(3e-4 if trig else 5e-5, 3.0 * e_cpu), "{}: HIP {:.2e} from the fp64 result, the fp32 oracle {:.2e}".format(what, e_hip, e_cpu)
        np.testing.assert_allclose(got[keep], ref[keep], rtol=5e-2, atol=2e-2, err_msg=what)
