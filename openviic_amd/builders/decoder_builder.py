"""``build_decoder(config, vocab)`` -- reference ``builders/decoder_builder.py:3-8``."""
from .registry import Registry

META_DECODER = Registry("DECODER_LAYER")


def build_decoder(config, vocab):
    return META_DECODER.get(config.ARCHITECTURE)(config, vocab)
