"""Attribute-style yaml configuration for the captioning engine.

The reference loads ``configs/*.yaml`` into a ``yacs.CfgNode`` and reads it only through
attribute access (reference ``configs/utils.py:4-5``; e.g. ``config.ENCODER.SELF_ATTENTION.HEAD``).
``yacs`` is not a dependency here: :class:`ConfigNode` gives the same attribute semantics on a
plain ``dict`` so the reference's yaml files load unchanged, plus dotted-key overrides for the
two keys the benchmark changes (``MODEL.DEVICE`` and ``MODEL.VISION_EMBEDDING.D_FEATURE``).
"""
from __future__ import annotations

import copy
from typing import Any, Dict, Iterable, Mapping, Optional, Tuple

import yaml


class ConfigNode(dict):
    """dict with recursive attribute access; a missing key raises ``AttributeError``."""

    def __init__(self, init_dict: Optional[Mapping[str, Any]] = None):
        super().__init__()
        for key, value in (init_dict or {}).items():
            self[key] = ConfigNode(value) if isinstance(value, Mapping) else value

    def __getattr__(self, name: str) -> Any:
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None

    def __setattr__(self, name: str, value: Any) -> None:
        self[name] = ConfigNode(value) if isinstance(value, Mapping) and not isinstance(value, ConfigNode) else value

    def __deepcopy__(self, memo):
        return ConfigNode({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def clone(self) -> "ConfigNode":
        return copy.deepcopy(self)

    def merge_from_list(self, overrides: Iterable[Tuple[str, Any]]) -> "ConfigNode":
        """Apply ``[("MODEL.DEVICE", "cpu"), ...]`` in place; intermediate nodes must exist."""
        for dotted, value in overrides:
            node = self
            parts = dotted.split(".")
            for part in parts[:-1]:
                if part not in node:
                    raise KeyError("config has no node '{}' (while setting '{}')".format(part, dotted))
                node = node[part]
            node[parts[-1]] = value
        return self

    def to_dict(self) -> Dict[str, Any]:
        return {k: (v.to_dict() if isinstance(v, ConfigNode) else v) for k, v in self.items()}


def get_config(yaml_file: str, overrides: Optional[Mapping[str, Any]] = None) -> ConfigNode:
    """Load a reference-format yaml; ``overrides`` maps dotted keys to values."""
    with open(yaml_file, "r") as f:
        node = ConfigNode(yaml.safe_load(f))
    if overrides:
        node.merge_from_list(overrides.items())
    return node


# ---------------------------------------------------------------------------------------------
# Programmatic equivalents of the reference's in-scope yaml files (MODEL section only), used by
# the tests and the benchmark on machines where the reference checkout is absent.
# ---------------------------------------------------------------------------------------------

def _attention(arch: str, d_model: int, heads: int, d_kv: int, d_ff: int, use_aoa: bool,
               stateful: bool, memory: Optional[int] = None) -> Dict[str, Any]:
    node = {
        "ARCHITECTURE": arch, "HEAD": heads, "D_MODEL": d_model, "D_KEY": d_kv, "D_VALUE": d_kv,
        "D_FF": d_ff, "USE_AOA": use_aoa, "CAN_BE_STATEFUL": stateful, "DROPOUT": 0.1,
    }
    if memory is not None:
        node["MEMORY"] = memory
    return node


_VARIANTS = {
    # name: (architecture, encoder, encoder attention, decoder, aoa)
    "standard_transformer": ("StandardTransformerUsingRegion", "Encoder", "ScaledDotProductAttention", "Decoder", False),
    "standard_transformer_using_region": ("StandardTransformerUsingRegion", "Encoder", "ScaledDotProductAttention", "Decoder", False),
    "standard_transformer_using_grid": ("StandardTransformerUsingGrid", "Encoder", "ScaledDotProductAttention", "Decoder", False),
    "attention_on_attention": ("StandardTransformerUsingRegion", "Encoder", "ScaledDotProductAttention", "Decoder", True),
    "meshed_memory_transformer": ("MeshedMemoryTransformer", "MultilevelEncoder", "AugmentedMemoryScaledDotProductAttention", "MeshedDecoder", False),
    "object_relation_transformer": ("ObjectRelationTransformer", "GeometricEncoder", "AugmentedGeometryScaledDotProductAttention", "Decoder", False),
}


def model_config(variant: str, *, d_feature: int = 2048, d_model: int = 512, heads: int = 8,
                 d_kv: int = 64, d_ff: int = 2048, layers: int = 3, memory: int = 40,
                 device: str = "cuda", trignometric_embedding: bool = False) -> ConfigNode:
    """Build the ``MODEL`` node of one of the in-scope reference configurations.

    Key names follow ``configs/standard_transformer.yaml:39-97``,
    ``configs/meshed_memory_transformer.yaml:38-97`` and
    ``configs/object_relation_transformer.yaml:39-95`` of the reference.
    """
    if variant not in _VARIANTS:
        raise KeyError("unknown model variant '{}' (have: {})".format(variant, ", ".join(sorted(_VARIANTS))))
    arch, encoder, enc_attention, decoder, aoa = _VARIANTS[variant]
    enc_self = _attention(enc_attention, d_model, heads, d_kv, d_ff, aoa, False,
                          memory if "Memory" in enc_attention else None)
    encoder_node: Dict[str, Any] = {"ARCHITECTURE": encoder, "D_MODEL": d_model, "LAYERS": layers,
                                    "SELF_ATTENTION": enc_self}
    if encoder == "GeometricEncoder":
        encoder_node["TRIGNOMETRIC_EMBEDDING"] = trignometric_embedding
    dec_attention: Dict[str, Any] = {
        "SELF_ATTENTION": _attention("ScaledDotProductAttention", d_model, heads, d_kv, d_ff, aoa, True),
        "ENC_ATTENTION": _attention("ScaledDotProductAttention", d_model, heads, d_kv, d_ff, aoa, False),
    }
    if decoder == "MeshedDecoder":
        dec_attention["N_ENCODER_LAYERS"] = layers
        dec_attention["D_MODEL"] = d_model
    return ConfigNode({
        "ARCHITECTURE": arch,
        "NAME": variant,
        "DEVICE": device,
        "VISION_EMBEDDING": {"ARCHITECTURE": "FeatureEmbedding", "D_FEATURE": d_feature,
                             "D_MODEL": d_model, "DROPOUT": 0.1},
        "ENCODER": encoder_node,
        "DECODER": {
            "ARCHITECTURE": decoder, "D_MODEL": d_model, "LAYERS": layers,
            "ATTENTION": dec_attention,
            "TEXT_EMBEDDING": {"ARCHITECTURE": "UsualEmbedding", "D_MODEL": d_model,
                               "D_EMBEDDING": 300, "WORD_EMBEDDING": None,
                               "WORD_EMBEDDING_CACHE": None, "DROPOUT": 0.1},
        },
    })


def dual_collaborative_config(*, d_region: int = 2048, d_grid: int = 2048, d_model: int = 512, heads: int = 8,
                              d_kv: int = 64, d_ff: int = 2048, layers: int = 3,
                              trignometric_embedding: bool = False):
    """``(VISION_EMBEDDING, ENCODER)`` nodes of the dual-collaborative (DLCT) encoder: the keys that
    ``GeometricDualFeatureEmbedding`` (reference ``vision_embeddings.py:47-55``) and
    ``DualCollaborativeLevelEncoder`` (``encoders.py:116-144``) read.  The reference ships no yaml for them."""
    att = _attention("AugmentedGeometryScaledDotProductAttention", d_model, heads, d_kv, d_ff, False, False)
    embedding = ConfigNode({"ARCHITECTURE": "GeometricDualFeatureEmbedding", "D_REGION_FEATURE": d_region,
                            "D_GRID_FEATURE": d_grid, "D_MODEL": d_model, "DROPOUT": 0.1})
    encoder = ConfigNode({"ARCHITECTURE": "DualCollaborativeLevelEncoder", "D_MODEL": d_model, "HEAD": heads,
                          "LAYERS": layers, "TRIGNOMETRIC_EMBEDDING": trignometric_embedding,
                          "SELF_ATTENTION": dict(att), "CROSS_ATTENTION": dict(att)})
    return embedding, encoder
