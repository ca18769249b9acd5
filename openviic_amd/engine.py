"""Python handle of the fused HIP engine (``ovc_encode`` / ``ovc_beam_search``).

``CaptionEngine`` reads the parameter tensors of a host-side model (``architectures.py``) into the
``ovc_model`` pointer table once, owns one cached device workspace per HIP stream and forwards
calls on the current stream -- independent batches issued on different streams overlap on the GPU.
Parameters are referenced, not copied: in-place updates of the tensors (``load_state_dict``, an
optimizer step) are seen by the engine; the one derived buffer (the geometric encoder's stacked
``fc_gs``) is refreshed from the live parameters on every call; re-allocation (``.to()``) drops the
engine (``BaseTransformer._apply``).

Numerics never depend on a timing: the order in which every GEMM sums over K is fixed per call site
(``csrc/gemm.hip``, K-order classes), the tiling measurement below only ranks bit-identical tilings.
"""
import ctypes
import json
import os
import threading

import torch

from . import native
from .native import check


# one tiling measurement at a time per process (it synchronises the stream and shares one scratch buffer); the tuning
# objective travels as an explicit argument of every tune / get / set call (ABI 6), not as process-wide state
_TUNE_LOCK = threading.Lock()


def _p(t):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise native.OvcError("engine parameters must be contiguous fp32 tensors on the HIP device "
                              "(got {} {} contiguous={})".format(t.device, t.dtype, t.is_contiguous()))
    return t.data_ptr()


def _lin(dst, linear, planes=None, mode=0):
    """Pointer pair of an nn.Linear; in a split-precision mode also the weight's pre-cut planes (``planes`` collects
    (weight, version, buffer, mode) so that a changed weight is cut again before the next call)."""
    dst.w = _p(linear.weight.detach())
    dst.b = _p(linear.bias.detach()) if linear.bias is not None else None
    if planes is not None and mode and linear.weight.shape[1] % 16 == 0:
        dst.planes = planes.add(linear.weight, mode)


class _WeightPlanes:
    """Pre-cut 16-bit planes of the GEMM weights for a split-precision engine (``ovc_split_weight``): one device buffer per
    weight, re-cut when the weight's version counter moved (optimizer step, ``load_state_dict``, ``copy_``)."""

    def __init__(self, lib):
        self.lib, self.entries, self.lock = lib, [], threading.Lock()

    def add(self, weight, mode):
        n, k = weight.shape
        buf = torch.empty(self.lib.ovc_split_weight_bytes(n, k, mode), dtype=torch.uint8, device=weight.device)
        self.entries.append([weight, -1, buf, mode])
        return buf.data_ptr()

    def refresh(self, force=False):
        """Cut again what changed (``force``: everything -- for weights modified behind the version counter, e.g. through
        ``.data``).  Other streams may decode with these buffers, so a re-cut ends with a stream synchronisation."""
        with self.lock:
            stale = [e for e in self.entries if force or e[0]._version != e[1]]
            for entry in stale:
                weight, _, buf, mode = entry
                n, k = weight.shape
                check(self.lib.ovc_split_weight(_p(weight.detach()), n, k, mode, buf.data_ptr(), native.stream_handle()),
                      "ovc_split_weight")
                entry[1] = weight._version
            if stale:
                torch.cuda.current_stream().synchronize()


def _norm(dst, ln):
    dst.g, dst.b = _p(ln.weight.detach()), _p(ln.bias.detach())


def _mha(dst, mha, keep, planes=None, mode=0):
    att = mha.attention
    for d, fc in ((dst.q, att.fc_q), (dst.k, att.fc_k), (dst.v, att.fc_v), (dst.o, att.fc_o)):
        _lin(d, fc, planes, mode)
    _norm(dst.ln, mha.layer_norm)
    if mha.use_aoa:
        _lin(dst.aoa_i, mha.informative_attention, planes, mode)
        _lin(dst.aoa_g, mha.gated_attention, planes, mode)
    if hasattr(att, "m_k"):
        dst.m_k, dst.m_v = _p(att.m_k.detach()), _p(att.m_v.detach())


def _ffn(dst, pwff, planes=None, mode=0):
    _lin(dst.fc1, pwff.fc1, planes, mode); _lin(dst.fc2, pwff.fc2, planes, mode); _norm(dst.ln, pwff.layer_norm)


class CaptionEngine:
    # measure GEMM tilings per shape on first use (one-off ~0.2 s, synchronises); OVC_AUTOTUNE=0 disables
    autotune = os.environ.get("OVC_AUTOTUNE", "1") != "0"
    # objective of the tiling measurement (ovc_gemm_tune_objective): 1 = isolated latency (default); c > 1 ranks
    # tilings by the time of c co-running copies, which picks larger tiles.  Measured with 4 batches in flight:
    # +0.6 % captions/s for c = 4 (same-box A/B), up to +3.7 % for a search under the real load
    # (round-1 probe, since removed); round 2, same box, alternating runs: c = 2 gives +2.5..3 % on 4 streams (22.0k ->
    # 22.7k) and -6 % on the single-stream kernel-scoped GEMM rate (91.5 -> 85.6 TFLOP/s).  Either way the bits are the same.
    tune_concurrency = int(os.environ.get("OVC_TUNE_CONCURRENCY", "1"))
    # replay the decode launch sequence as a hipGraph from the third call of a shape on (OVC_GRAPH=0: plain launches)
    use_graph = os.environ.get("OVC_GRAPH", "1") != "0"
    # pad the region axis to a multiple of this with zero rows before decoding (1 = exact shapes).  Results are
    # identical; with ragged real-data batches a bucket of 8 or 16 bounds the number of distinct shapes (graphs).
    region_bucket = int(os.environ.get("OVC_REGION_BUCKET", "1"))
    # GEMM arithmetic: "f32" = fp32 MFMA, the parity mode (default, the only mode the headline numbers use).  Opt-in,
    # uncredited split precision: "bf16x6" cuts every GEMM's fp32 operands into three bf16 planes and contracts them on
    # the 16-bit matrix path with fp32 accumulation (6 plane products); "f16x3" uses two fp16 planes (scaled residual,
    # 3 products, 22 bits per operand; operands beyond fp16's range saturate) -- faster, fp32 in and out, but NOT
    # bit-identical to "f32" (DESIGN.md section 5a).  The one- and two-plane bf16 modes of round 2 failed the parity bar
    # and were removed.
    PRECISIONS = {"f32": 0, "bf16x6": 3, "f16x3": 4}
    # split-precision modes: cut every GEMM weight into its planes once (and again when it changes) instead of in every
    # workgroup of every launch -- same bits, W then bypasses conversion and LDS (OVC_PRECUT_WEIGHTS=0: A/B switch)
    precut_weights = os.environ.get("OVC_PRECUT_WEIGHTS", "1") != "0"
    precision = os.environ.get("OVC_PRECISION", "f32")
    # stop issuing decode steps once every beam of every image has ended (ovc_beam_search_early: identical results, but the call
    # blocks the host thread until the search is one step from its end -- use one host thread per stream to overlap batches).
    # Pays for real captions (they end well before max_len); random-weight benchmarks never emit <eos>.  Per call:
    # beam_search(..., early_exit=True).
    early_exit = os.environ.get("OVC_EARLY_EXIT", "0") != "0"

    def __init__(self, model, tune_concurrency=None, precision=None):
        self.lib = native.load()
        self.model = model
        if precision is not None:
            self.precision = precision
        if self.precision not in self.PRECISIONS:
            raise native.OvcError("precision must be one of {} (got {!r})".format(sorted(self.PRECISIONS), self.precision))
        if tune_concurrency is not None:          # per engine: e.g. 2 for a host that alternates batches over 3-4 streams
            self.tune_concurrency = int(tune_concurrency)
        self._keep = []          # tensors created here whose storage the pointer table references
        self._fc_g = None
        # split-precision modes: the GEMM weights pre-cut into 16-bit planes (read straight from memory by the kernels)
        self._planes = _WeightPlanes(self.lib) if self.precision != "f32" and self.precut_weights else None
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
        mode = self.PRECISIONS[self.precision]
        pl = self._planes
        _lin(d.proj, model.vision_embedding.proj, pl, 3 if mode == 4 else mode)   # f16x3: the features go through bf16 planes
        _norm(d.enc_ln, enc.layer_norm)
        if d.enc_kind == native.ENC_GEOMETRIC:
            d.trig, d.d_g = int(bool(enc.trignometric_embedding)), enc.d_g
            # the box-relation kernel wants the per-head Linear(d_g, 1) layers stacked: the only parameters the
            # engine holds as a COPY, re-filled from the live tensors by _refresh_derived() on every call
            w = torch.cat([fc.weight.detach() for fc in enc.fc_gs], dim=0).contiguous()
            b = torch.cat([fc.bias.detach() for fc in enc.fc_gs], dim=0).contiguous()
            self._fc_g = (w, b)
            self._keep += [w, b]
            d.fc_g_w, d.fc_g_b = _p(w), _p(b)
        for i, layer in enumerate(enc.layers):
            _mha(d.enc[i].att, layer.mhatt, self._keep, pl, mode)
            _ffn(d.enc[i].ffn, layer.pwff, pl, mode)
        for i, layer in enumerate(dec.layers):
            _mha(d.dec[i].self_att, layer.self_attn, self._keep, pl, mode)
            _mha(d.dec[i].cross_att, layer.enc_attn, self._keep, pl, mode)
            _ffn(d.dec[i].ffn, layer.pwff, pl, mode)
            if d.dec_kind == native.DEC_MESHED:
                for j, fc in enumerate(layer.fc_alphas):
                    _lin(d.dec[i].alpha[j], fc, pl, mode)
        d.word_emb = _p(dec.word_emb.components.weight.detach())
        d.pos_emb = _p(dec.pos_emb.weight.detach())
        d.fc = _p(dec.fc.weight.detach())
        if pl is not None and dec.fc.weight.shape[1] % 16 == 0:
            d.fc_planes = pl.add(dec.fc.weight, mode)
        d.tune_objective = max(1, min(8, int(self.tune_concurrency)))
        d.precision = self.PRECISIONS[self.precision]
        if self.precision == "f16x3":
            # fp16 planes: an operand beyond fp16's range would turn into inf; the weights can be checked here, once
            top = max(float(p.detach().abs().max()) for p in model.parameters() if p.numel())
            if not top < 65504.0:
                raise native.OvcError("precision='f16x3' needs every weight inside fp16's range (largest |w| = {:g}); "
                                      "use 'bf16x6' or the default 'f32'".format(top))
        return d

    def _refresh_derived(self):
        """Re-fill the stacked ``fc_gs`` copy from the live parameters (h * d_g floats, on the current stream): a
        ``load_state_dict`` or an optimizer step between two calls must not leave the engine on stale weights."""
        if self._fc_g is not None:
            enc = self.model.encoder
            torch.cat([fc.weight.detach() for fc in enc.fc_gs], dim=0, out=self._fc_g[0])
            torch.cat([fc.bias.detach() for fc in enc.fc_gs], dim=0, out=self._fc_g[1])
        if self._planes is not None:
            self._planes.refresh()

    def recut_weights(self):
        """Split-precision modes: rebuild every weight's pre-cut planes (only needed after modifying weights in a way that
        does not move their version counter, e.g. through ``.data``)."""
        if self._planes is not None:
            self._planes.refresh(force=True)

    # -- GEMM tiling selection ------------------------------------------------------------------
    def gemm_shapes(self, B, N, k):
        """(M, seg_n, nseg, K, kchains, ksplit, epilogue) of every GEMM the engine issues for batch B, N regions, beam k --
        enumerated by the library itself (``ovc_engine_gemm_shapes`` walks the real launch sequence in dry mode)."""
        cap = 64
        while True:
            buf = (ctypes.c_int32 * (7 * cap))()
            n = self.lib.ovc_engine_gemm_shapes(ctypes.byref(self.desc), B, N, k, buf, cap)
            check(min(n, 0), "ovc_engine_gemm_shapes")
            if n <= cap:
                return sorted(tuple(buf[7 * i + j] for j in range(7)) for i in range(n))
            cap = n

    def tune(self, B, N, k):
        """Time the GEMM tilings of each of the engine's (shape, K-order class) once and let the library remember the
        fastest (synchronises; every tiling of the class runs 2 + 3 x 6 launches per shape: ~0.5 s for the BASELINE model's
        shapes).  Speed only: all tilings of a class give the same bits.  Shapes for which the
        library already holds a MEASURED entry with M within a factor of two (another region count or batch size; for the
        transposed vocabulary product, whose batch size is its column count: seg_n within a factor of two) borrow its
        choice and are not measured, so batches whose N varies inside such a range never wait here after the first one (a
        shape that borrowed leaves no entry of its own: a later shape beyond the factor of two of every measured M is measured
        once more)."""
        objective = int(self.desc.tune_objective)
        key = (B, N, k)
        if key in self._tuned:
            return
        with _TUNE_LOCK:
            self._tune_locked(B, N, k, objective)
        self._tuned.add(key)

    def _tune_locked(self, B, N, k, objective):
        shapes = self.gemm_shapes(B, N, k)
        # each objective has its own table in the library, named explicitly in every call below
        cache_path = os.environ.get("OVC_TUNE_CACHE")       # optional json: {"M,seg_n,nseg,K,kchains,ksplit@objective": tiling}
        cache = {}
        if cache_path and os.path.exists(cache_path):
            with open(cache_path) as f:
                cache = json.load(f)
            for shape in shapes:
                name = ",".join(map(str, shape)) + "@%d" % objective
                if name in cache:
                    self.lib.ovc_gemm_tuned_set(*shape[:6], objective, int(cache[name]))
        todo = [sh for sh in shapes if self.lib.ovc_gemm_tuned_get(*sh[:6], objective, 1) < 0]
        if todo:
            # operands + output (K-split shapes: one partial output per slice)
            # (+ room for pre-cut weight planes in the split-precision classes: the tuner then ranks the instances the engine runs)
            # (+ room for the log-softmax block pieces of a wide decode-class product: it is then ranked with that epilogue)
            need = max(4 * (m * kk + sn * ns * kk + m * sn * ns * ks) + 256 +
                       (8 * m * (sn // 32 + 4) if ep == 1 else 8 * sn * (m // 32 + 4) if ep == 2 else 0) +
                       (ns * self.lib.ovc_split_weight_bytes(sn, kk, kc - 100) if kc > 100 else 0) for m, sn, ns, kk, kc, ks, ep in todo)
            scratch = torch.empty(need // 4 + 16, dtype=torch.float32, device=self.device).normal_()
            for sh in todo:
                check(self.lib.ovc_gemm_tune(*sh[:6], objective, sh[6], scratch.data_ptr(), scratch.numel() * 4, native.stream_handle()),
                      "ovc_gemm_tune{}".format(sh))
            torch.cuda.current_stream().synchronize()
        if cache_path and todo:
            for shape in shapes:
                t = self.lib.ovc_gemm_tuned_get(*shape[:6], objective, 0)
                if t >= 0:
                    cache[",".join(map(str, shape)) + "@%d" % objective] = t
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
            old = self._workspaces.pop(key, None)
            if old is not None:
                # captured graphs reference the old buffer's addresses: drop them before it is freed
                self.lib.ovc_graph_cache_drop_workspace(old.data_ptr())
            # grow geometrically so that a slowly increasing region count does not re-allocate (and re-capture) every time
            size = need if old is None else max(need, int(old.numel() * 1.25))
            ws = self._workspaces[key] = torch.empty(size, dtype=torch.uint8, device=self.device)
        return ws, need

    def release(self):
        """Drop this engine's workspaces and the hipGraphs captured on them."""
        lib = getattr(self, "lib", None)
        for ws in getattr(self, "_workspaces", {}).values():
            if lib is not None:
                lib.ovc_graph_cache_drop_workspace(ws.data_ptr())
        self._workspaces = {}

    def __del__(self):
        try:
            self.release()
        except Exception:       # interpreter shutdown: the library or torch may already be gone
            pass

    def _bucketed(self, features, boxes):
        """Pad the region axis up to a multiple of ``region_bucket`` with all-zero rows.  Exact: a zero feature row IS
        the reference's padding (``utils/instance.py:156-171`` pads ragged batches the same way, ``models/utils.py:48-61``
        masks such rows as keys, positions are indexed by region), and every GEMM sums K in a shape-independent order.
        (Encoders with memory slots: the slots follow the regions in the key order, so padding moves them to other
        accumulator registers -- the same math in another summation order, equal up to rounding, not bit for bit.)
        Fewer distinct N means fewer captured graphs when the region count varies from batch to batch."""
        bucket = max(1, int(self.region_bucket))
        N = features.shape[1]
        target = -(-N // bucket) * bucket
        if target > native.OVC_MAX_REGIONS:   # the bucket would pass the engine's region limit: keep the exact shape (an N beyond
            target = N                        # the limit then reaches ovc_workspace_bytes and raises -- padding never crops)
        # Up to 128 regions (and 192 keys: regions + the encoder's memory slots) the attention kernels keep a query's scores in
        # registers, beyond that the keys pass in tiles under an online softmax (csrc/attention.hip): the two forms round
        # differently, so a bucket never carries a batch across that edge -- the padded decode stays bit-identical to the
        # unpadded one.
        memory = int(getattr(getattr(self, "desc", None), "memory", 0) or 0)
        for edge in sorted({max(192 - memory, 0), 128}):
            if N <= edge < target:
                target = edge
                break
        if target == N:
            return features, boxes
        pad = target - N
        features = torch.nn.functional.pad(features, (0, 0, 0, pad))
        if boxes is not None:
            boxes = torch.nn.functional.pad(boxes, (0, 0, 0, pad))
        return features, boxes

    @staticmethod
    def _features(x, name):
        if not x.is_cuda or x.dtype != torch.float32:
            raise native.OvcError("{} must be an fp32 tensor on the HIP device (got {} {})".format(name, x.device, x.dtype))
        return x.contiguous()

    # -- calls --------------------------------------------------------------------------------
    def encode(self, features, boxes=None):
        features, boxes = self._checked_inputs(features, boxes)
        B, N = features.shape[:2]
        self._refresh_derived()
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

    def beam_search(self, features, boxes, batch_size, beam_size, out_size=1, return_probs=False, early_exit=None):
        """``early_exit`` (default: the class attribute / OVC_EARLY_EXIT): see above; ``self.last_steps_run`` then holds the
        number of decode steps that were issued for the call."""
        features, boxes = self._checked_inputs(features, boxes)
        features, boxes = self._bucketed(features, boxes)
        B, N = features.shape[:2]
        if B != batch_size:
            raise native.OvcError("batch_size={} but features hold {} images".format(batch_size, B))
        self._refresh_derived()
        d = self.desc
        T, V = d.max_len, d.vocab
        if self.autotune:
            self.tune(B, N, beam_size)
        ws, need = self._get_workspace(B, N, beam_size, return_probs)
        ids = torch.empty(B, out_size, T, dtype=torch.int64, device=self.device)
        logp = torch.empty(B, out_size, T, dtype=torch.float32, device=self.device)
        everything = torch.empty(B, beam_size, T, V, dtype=torch.float32, device=self.device) if return_probs else None
        early = self.early_exit if early_exit is None else bool(early_exit)
        self.last_steps_run = T
        if early and not return_probs:
            steps = ctypes.c_int(0)
            check(self.lib.ovc_beam_search_early(ctypes.byref(d), features.data_ptr(),
                                                 None if boxes is None else boxes.data_ptr(), B, N, beam_size, out_size,
                                                 ws.data_ptr(), need, ids.data_ptr(), logp.data_ptr(), ctypes.byref(steps),
                                                 native.stream_handle()), "ovc_beam_search_early")
            self.last_steps_run = steps.value
        elif self.use_graph and not return_probs:
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
