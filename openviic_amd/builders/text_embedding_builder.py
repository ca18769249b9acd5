"""``build_text_embedding(config, vocab)`` -- reference ``builders/text_embedding_builder.py:3-8``."""
from .registry import Registry

META_TEXT_EMBEDDING = Registry("TEXT_EMBEDDING")


def build_text_embedding(config, vocab):
    return META_TEXT_EMBEDDING.get(config.ARCHITECTURE)(config, vocab)
