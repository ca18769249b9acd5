"""Python handle of the fused HIP engine (``ovc_encode`` / ``ovc_beam_search``).

``CaptionEngine`` reads the parameter tensors of a host-side model (``architectures.py``) into the
``ovc_model`` pointer table once, owns one cached device workspace per HIP stream and forwards
calls on the current stream -- independent batches issued on different streams overlap on the GPU.  Parameters are referenced, not copied: in-place updates of the tensors are seen by
the engine; re-allocation (``.to()``) drops the engine (``BaseTransformer._apply``).
"""
import ctypes
import json
import os

import torch

from . import native
from .native import check


def _p(t):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise native.OvcError("engine parameters must be contiguous fp32 tensors on the HIP device "
                              "(got {} {} contiguous={})".format(t.device, t.dtype, t.is_contiguous()))
    return t.data_ptr()


def _lin(dst, linear):
    dst.w = _p(linear.weight.detach())
    dst.b = _p(linear.bias.detach()) if linear.bias is not None else None


def _norm(dst, ln):
    dst.g, dst.b = _p(ln.weight.detach()), _p(ln.bias.detach())


def _mha(dst, mha, keep):
    att = mha.attention
    _lin(dst.q, att.fc_q); _lin(dst.k, att.fc_k); _lin(dst.v, att.fc_v); _lin(dst.o, att.fc_o)
    _norm(dst.ln, mha.layer_norm)
    if mha.use_aoa:
        _lin(dst.aoa_i, mha.informative_attention)
        _lin(dst.aoa_g, mha.gated_attention)
    if hasattr(att, "m_k"):
        dst.m_k, dst.m_v = _p(att.m_k.detach()), _p(att.m_v.detach())


def _ffn(dst, pwff):
    _lin(dst.fc1, pwff.fc1); _lin(dst.fc2, pwff.fc2); _norm(dst.ln, pwff.layer_norm)


class CaptionEngine:
    # measure GEMM tilings per shape on first use (one-off ~0.2 s, synchronises); OVC_AUTOTUNE=0 disables
    autotune = os.environ.get("OVC_AUTOTUNE", "1") != "0"
    # objective of the tiling measurement (ovc_gemm_tune_objective): 1 = isolated latency (default); c > 1 ranks
    # tilings by the time of c co-running copies, which picks larger tiles.  Measured with 4 batches in flight:
    # +0.6 % captions/s for c = 4 (same-box A/B), up to +3.7 % for a search under the real load
    # (tools/tiling_throughput_probe.py), while the same kernels run alone drop from 91 to 69 TFLOP/s.
    tune_concurrency = int(os.environ.get("OVC_TUNE_CONCURRENCY", "1"))
    # replay the decode launch sequence as a hipGraph from the third call of a shape on (OVC_GRAPH=0: plain launches)
    use_graph = os.environ.get("OVC_GRAPH", "1") != "0"

    def __init__(self, model):
        self.lib = native.load()
        self.model = model
        self._keep = []          # tensors created here whose storage the pointer table references
        self.desc = self._describe(model)
        self._workspaces = {}    # one scratch buffer per HIP stream: concurrent batches never share state
        self._tuned = set()
        self.device = next(model.parameters()).device
        if self.device.type != "cuda":
            raise native.OvcError("the fused engine needs the model on a HIP device (got {}); "
                                  "there is no CPU path".format(self.device))

    # -- pointer table ------------------------------------------------------------------------
    def _describe(self, model) -> native.Model:
        from .modules import encoders, decoders
        d = native.Model()
        enc, dec = model.encoder, model.decoder
        first = enc.layers[0].mhatt.attention
        d.abi = native.ABI_VERSION
        d.enc_kind = (native.ENC_MULTILEVEL if isinstance(enc, encoders.MultilevelEncoder)
                      else native.ENC_GEOMETRIC if isinstance(enc, encoders.GeometricEncoder)
                      else native.ENC_PLAIN)
        d.dec_kind = native.DEC_MESHED if isinstance(dec, decoders.MeshedDecoder) else native.DEC_PLAIN
        d.d_feat = model.vision_embedding.proj.in_features
        d.d_model, d.heads, d.d_k, d.d_v = first.d_model, first.h, first.d_k, first.d_v
        d.d_ff = enc.layers[0].pwff.fc1.out_features
        d.n_enc, d.n_dec = len(enc.layers), len(dec.layers)
        d.n_levels = dec.layers[0].nlayers if d.dec_kind == native.DEC_MESHED else 1
        d.memory = getattr(first, "m", 0)
        d.vocab, d.max_len = dec.fc.out_features, dec.max_len
        d.pad_idx, d.bos_idx, d.eos_idx = dec.padding_idx, model.vocab.bos_idx, model.eos_idx
        d.ln_eps = enc.layer_norm.eps
        if len(enc.layers) > native.OVC_MAX_LAYERS or len(dec.layers) > native.OVC_MAX_LAYERS:
            raise native.OvcError("at most {} layers are supported".format(native.OVC_MAX_LAYERS))
        _lin(d.proj, model.vision_embedding.proj)
        _norm(d.enc_ln, enc.layer_norm)
        if d.enc_kind == native.ENC_GEOMETRIC:
            d.trig, d.d_g = int(bool(enc.trignometric_embedding)), enc.d_g
            w = torch.cat([fc.weight.detach() for fc in enc.fc_gs], dim=0).contiguous()
            b = torch.cat([fc.bias.detach() for fc in enc.fc_gs], dim=0).contiguous()
            self._keep += [w, b]
            d.fc_g_w, d.fc_g_b = _p(w), _p(b)
        for i, layer in enumerate(enc.layers):
            _mha(d.enc[i].att, layer.mhatt, self._keep)
            _ffn(d.enc[i].ffn, layer.pwff)
        for i, layer in enumerate(dec.layers):
            _mha(d.dec[i].self_att, layer.self_attn, self._keep)
            _mha(d.dec[i].cross_att, layer.enc_attn, self._keep)
            _ffn(d.dec[i].ffn, layer.pwff)
            if d.dec_kind == native.DEC_MESHED:
                for j, fc in enumerate(layer.fc_alphas):
                    _lin(d.dec[i].alpha[j], fc)
        d.word_emb = _p(dec.word_emb.components.weight.detach())
        d.pos_emb = _p(dec.pos_emb.weight.detach())
        d.fc = _p(dec.fc.weight.detach())
        return d

    # -- GEMM tiling selection ------------------------------------------------------------------
    def gemm_shapes(self, B, N, k):
        """(M, seg_n, nseg, K) of every GEMM the engine issues for batch B, N regions, beam k."""
        d = self.desc
        dm, hk, hv, ff, L = d.d_model, d.heads * d.d_k, d.heads * d.d_v, d.d_ff, d.n_dec
        shapes = set()
        bn = B * N
        shapes.add((bn, dm, 1, d.d_feat))
        shapes.update({(bn, hk, 3, dm), (bn, dm, 1, hv), (bn, ff, 1, dm), (bn, dm, 1, ff)})
        for l0 in range(0, L, 4):
            shapes.add((bn, hk, 2 * min(4, L - l0), dm))
        for rows in {B, B * k}:
            shapes.update({(rows, hk, 3, dm), (rows, dm, 1, hv), (rows, hk, 1, dm), (rows, ff, 1, dm),
                           (rows, dm, 1, ff), (rows, d.vocab, 1, dm)})
            if d.dec_kind == native.DEC_MESHED:
                shapes.add((rows, dm, d.n_levels if dm % 64 == 0 else 1, 2 * dm))      # level gates, one segment per level
                shapes.add((rows * d.n_levels, dm, 1, hv))           # shared output projection over the stacked levels
        for rows, aoa in ((bn, self.model.encoder.layers[0].mhatt.use_aoa),
                          (B, self.model.decoder.layers[0].self_attn.use_aoa),
                          (B * k, self.model.decoder.layers[0].self_attn.use_aoa)):
            if aoa:
                shapes.add((rows, dm, 2 if dm % 64 == 0 else 1, 2 * dm))
        return sorted(shapes)

    def tune(self, B, N, k):
        """Time every GEMM tiling on the engine's shapes once (synchronises; ~0.2 s) and let the
        library remember the fastest per shape."""
        key = (B, N, k)
        if key in self._tuned:
            return
        shapes = self.gemm_shapes(B, N, k)
        objective = max(1, min(8, int(self.tune_concurrency)))
        cache_path = os.environ.get("OVC_TUNE_CACHE")       # optional json: {"M,seg_n,nseg,K@objective": ovc_gemm_tuned_get code}
        cache = {}
        if cache_path and os.path.exists(cache_path):
            with open(cache_path) as f:
                cache = json.load(f)
            for shape in shapes:
                name = ",".join(map(str, shape)) + "@%d" % objective
                if name in cache:
                    self.lib.ovc_gemm_tuned_set(*shape, int(cache[name]))
        check(self.lib.ovc_gemm_tune_objective(objective), "ovc_gemm_tune_objective")
        # operands + output; single-segment shapes also hold the partial outputs of a 4-way K split
        need = max(4 * (m * kk + sn * ns * kk + m * sn * ns * (4 if ns == 1 and m * sn < 4 << 20 else 1)) + 256
                   for m, sn, ns, kk in shapes)
        scratch = torch.empty(need // 4 + 16, dtype=torch.float32, device=self.device).normal_()
        for m, sn, ns, kk in shapes:
            check(self.lib.ovc_gemm_tune(m, sn, ns, kk, scratch.data_ptr(), scratch.numel() * 4, native.stream_handle()),
                  "ovc_gemm_tune{}".format((m, sn, ns, kk)))
        torch.cuda.current_stream().synchronize()
        self._tuned.add(key)
        if cache_path:
            for shape in shapes:
                cache[",".join(map(str, shape)) + "@%d" % objective] = self.lib.ovc_gemm_tuned_get(*shape)
            os.makedirs(os.path.dirname(os.path.abspath(cache_path)), exist_ok=True)
            with open(cache_path, "w") as f:
                json.dump(cache, f, indent=0, sort_keys=True)

    # -- workspace ----------------------------------------------------------------------------
    def _get_workspace(self, B, N, k, return_probs):
        need = self.lib.ovc_workspace_bytes(ctypes.byref(self.desc), B, N, k, 1 if return_probs else 0)
        if need == 0:
            raise native.OvcError("unsupported engine configuration (B={}, N={}, beam={}; see ovc_workspace_bytes)"
                                  .format(B, N, k))
        key = torch.cuda.current_stream().cuda_stream
        ws = self._workspaces.get(key)
        if ws is None or ws.numel() < need:
            self._workspaces.pop(key, None)
            ws = self._workspaces[key] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return ws, need

    @staticmethod
    def _features(x, name):
        if not x.is_cuda or x.dtype != torch.float32:
            raise native.OvcError("{} must be an fp32 tensor on the HIP device (got {} {})".format(name, x.device, x.dtype))
        return x.contiguous()

    # -- calls --------------------------------------------------------------------------------
    def encode(self, features, boxes=None):
        features, boxes = self._checked_inputs(features, boxes)
        B, N = features.shape[:2]
        ws, need = self._get_workspace(B, N, 1, False)
        d = self.desc
        shape = (B, d.n_levels, N, d.d_model) if d.enc_kind == native.ENC_MULTILEVEL else (B, N, d.d_model)
        out = torch.empty(shape, dtype=torch.float32, device=self.device)
        mask = torch.empty(B, N, dtype=torch.uint8, device=self.device)
        check(self.lib.ovc_encode(ctypes.byref(d), features.data_ptr(), None if boxes is None else boxes.data_ptr(),
                                  B, N, ws.data_ptr(), need, out.data_ptr(), mask.data_ptr(),
                                  native.stream_handle()), "ovc_encode")
        return out, mask.view(torch.bool)[:, None, None, :]

    def _checked_inputs(self, features, boxes):
        """The kernels index with the model's strides: shapes are verified here, on the host, before any launch."""
        features = self._features(features, "features")
        boxes = None if boxes is None else self._features(boxes, "boxes")
        if features.dim() != 3 or features.shape[2] != self.desc.d_feat:
            raise native.OvcError("features must be (B, N, {}); got {}".format(self.desc.d_feat, tuple(features.shape)))
        if boxes is not None and tuple(boxes.shape) != (features.shape[0], features.shape[1], 4):
            raise native.OvcError("boxes must be {}; got {}".format((features.shape[0], features.shape[1], 4),
                                                                    tuple(boxes.shape)))
        if self.desc.enc_kind == native.ENC_GEOMETRIC and boxes is None:
            raise native.OvcError("the geometric encoder needs region boxes")
        return features, boxes

    def beam_search(self, features, boxes, batch_size, beam_size, out_size=1, return_probs=False):
        features, boxes = self._checked_inputs(features, boxes)
        B, N = features.shape[:2]
        if B != batch_size:
            raise native.OvcError("batch_size={} but features hold {} images".format(batch_size, B))
        d = self.desc
        T, V = d.max_len, d.vocab
        if self.autotune:
            self.tune(B, N, beam_size)
        ws, need = self._get_workspace(B, N, beam_size, return_probs)
        ids = torch.empty(B, out_size, T, dtype=torch.int64, device=self.device)
        logp = torch.empty(B, out_size, T, dtype=torch.float32, device=self.device)
        everything = torch.empty(B, beam_size, T, V, dtype=torch.float32, device=self.device) if return_probs else None
        if self.use_graph and not return_probs:
            check(self.lib.ovc_beam_search_graph(ctypes.byref(d), features.data_ptr(),
                                                 None if boxes is None else boxes.data_ptr(), B, N, beam_size, out_size,
                                                 ws.data_ptr(), need, ids.data_ptr(), logp.data_ptr(),
                                                 native.stream_handle()), "ovc_beam_search_graph")
        else:
            check(self.lib.ovc_beam_search(ctypes.byref(d), features.data_ptr(), None if boxes is None else boxes.data_ptr(),
                                           B, N, beam_size, out_size, ws.data_ptr(), need, ids.data_ptr(), logp.data_ptr(),
                                           None if everything is None else everything.data_ptr(),
                                           native.stream_handle()), "ovc_beam_search")
        if out_size == 1:
            ids, logp = ids.squeeze(1), logp.squeeze(1)
        return (ids, logp, everything) if return_probs else (ids, logp)
