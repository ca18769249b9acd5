"""Model architectures (reference ``models/base_transformer.py:8-53``, ``standard_stransformer.py``,
``meshed_memory_transformer.py``, ``object_relation_transformer.py``), registered under the
reference's names so its yaml files resolve unchanged.

API kept: ``forward(input_features) -> log-probs (B,T,V)``; ``encoder_forward(input_features) ->
(encoder_features, padding_mask (B,1,1,N) bool)``; ``step(t, prev_output)``;
``beam_search(input_features, batch_size, beam_size, out_size=1, return_probs=False)``.

``beam_search`` is the accelerated path: one call into the fused HIP engine
(``csrc/engine.hip``) which runs encoder, every decode step and the beam bookkeeping on the
device without host round trips.  It needs the HIP library and a GPU; there is no CPU fallback.
"""
import torch

from . import engine
from .builders.decoder_builder import build_decoder
from .builders.encoder_builder import build_encoder
from .builders.model_builder import META_ARCHITECTURE
from .builders.vision_embedding_builder import build_vision_embedding
from .modules.beam_search import BeamSearch
from .modules.containers import Module


class BaseTransformer(Module):
    feature_field = "region_features"
    uses_boxes = False

    def __init__(self, config, vocab):
        super().__init__()
        self.vocab = vocab
        self.max_len = vocab.max_caption_length
        self.eos_idx = vocab.eos_idx
        self.register_state("encoder_features", None)
        self.register_state("encoder_padding_mask", None)
        self.device = torch.device(config.DEVICE)
        self.vision_embedding = build_vision_embedding(config.VISION_EMBEDDING)
        self.encoder = build_encoder(config.ENCODER)
        self.decoder = build_decoder(config.DECODER, vocab)
        self._engine = None
        self._predict_pipeline = None

    def init_weights(self):
        for p in self.parameters():
            if p.dim() > 1:
                torch.nn.init.xavier_uniform_(p)

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._engine = None                       # parameter storage may have moved
        self._predict_pipeline = None             # (data.predict_feature_files: streams and pinned buffers of the old device)
        if any(True for _ in self.parameters()):
            self.device = next(self.parameters()).device
        return out

    # -- operator-by-operator API ---------------------------------------------------------
    def encoder_forward(self, input_features):
        features, padding_mask = self.vision_embedding(input_features[self.feature_field])
        if self.uses_boxes:
            # The reference passes a single Instance here, which its GeometricEncoder.forward
            # (features, boxes, padding_mask) rejects with a TypeError
            # (object_relation_transformer.py:38-42 vs encoders.py:93); wired by keyword instead.
            out = self.encoder(features=features, boxes=input_features["region_boxes"], padding_mask=padding_mask)
        else:
            out = self.encoder(features=features, padding_mask=padding_mask)
        return out, padding_mask

    def forward(self, input_features):
        encoder_features, encoder_padding_mask = self.encoder_forward(input_features)
        return self.decoder(caption_tokens=input_features["caption_tokens"],
                            encoder_features=encoder_features,
                            encoder_attention_mask=encoder_padding_mask)

    def step(self, t, prev_output):
        bs = self.encoder_features.shape[0]
        if t == 0:
            it = torch.full((bs, 1), self.vocab.bos_idx, dtype=torch.long, device=self.encoder_features.device)
        else:
            it = prev_output
        return self.decoder(caption_tokens=it, encoder_features=self.encoder_features,
                            encoder_attention_mask=self.encoder_padding_mask)

    # -- accelerated path -------------------------------------------------------------------
    def beam_search(self, input_features, batch_size: int, beam_size: int, out_size=1, return_probs=False,
                    fused=True, **kwargs):
        """Beam-search decode (``base_transformer.py:45-53`` + ``beam_search.py:85-118``).

        ``fused=True`` (default) runs the whole search in the HIP engine.  ``fused=False`` runs the
        reference's host loop (``modules/beam_search.py``) over the step-wise API -- every operator
        still native -- and exists for parity checks of ``step`` / ``statefulness``.
        """
        if fused:
            if self._engine is None:
                self._engine = engine.CaptionEngine(self)
            boxes = input_features["region_boxes"] if self.uses_boxes else None
            return self._engine.beam_search(input_features[self.feature_field], boxes, batch_size, beam_size,
                                            out_size=out_size, return_probs=return_probs, early_exit=kwargs.get("early_exit"))
        searcher = BeamSearch(model=self, max_len=self.max_len, eos_idx=self.eos_idx, beam_size=beam_size,
                              b_s=batch_size, device=self.device)
        with self.statefulness(batch_size):
            self.encoder_features, self.encoder_padding_mask = self.encoder_forward(input_features)
            return searcher.apply(out_size, return_probs, **kwargs)


@META_ARCHITECTURE.register()
class StandardTransformerUsingRegion(BaseTransformer):
    feature_field = "region_features"


@META_ARCHITECTURE.register()
class StandardTransformerUsingGrid(BaseTransformer):
    feature_field = "grid_features"


@META_ARCHITECTURE.register()
class MeshedMemoryTransformer(BaseTransformer):
    feature_field = "region_features"


@META_ARCHITECTURE.register()
class ObjectRelationTransformer(BaseTransformer):
    feature_field = "region_features"
    uses_boxes = True
