"""``build_vision_embedding(config)`` -- reference ``builders/vision_embedding_builder.py:3-8``."""
from .registry import Registry

META_VISION_EMBEDDING = Registry("META_VISION_EMBEDDING")


def build_vision_embedding(config):
    return META_VISION_EMBEDDING.get(config.ARCHITECTURE)(config)
