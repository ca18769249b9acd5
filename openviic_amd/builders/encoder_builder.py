"""``build_encoder(config)`` -- reference ``builders/encoder_builder.py:3-8``."""
from .registry import Registry

META_ENCODER = Registry("ENCODER_LAYER")


def build_encoder(config):
    return META_ENCODER.get(config.ARCHITECTURE)(config)
