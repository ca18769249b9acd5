"""Registry / factory surface of the engine (same names as the reference's ``builders`` package).

Importing this package imports the module classes so that every registry is populated, as the
reference does in ``builders/__init__.py:1-2``.
"""
from .registry import Registry
from .factories import (META_ARCHITECTURE, META_ATTENTION, META_DECODER, META_ENCODER, META_TEXT_EMBEDDING,
                        META_VISION_EMBEDDING, build_attention, build_decoder, build_encoder, build_model,
                        build_text_embedding, build_vision_embedding)

from .. import modules as _modules            # noqa: F401  (registers module classes)
from .. import architectures as _architectures  # noqa: F401  (registers model classes)

__all__ = [
    "Registry",
    "META_ATTENTION", "build_attention", "META_ENCODER", "build_encoder",
    "META_DECODER", "build_decoder", "META_VISION_EMBEDDING", "build_vision_embedding",
    "META_TEXT_EMBEDDING", "build_text_embedding", "META_ARCHITECTURE", "build_model",
]
