"""All registries and ``build_*`` factories of the engine in one place.

The reference spreads these over one five-line module per registry (``builders/*_builder.py``);
the per-name modules of this package re-export from here so that the reference's import paths
(``builders.model_builder.build_model`` ...) keep working.  A factory resolves
``config.ARCHITECTURE`` in its registry and calls the class with ``(config)`` or ``(config, vocab)``.
"""
import torch

from .registry import Registry


def _factory(registry: Registry, takes_vocab: bool):
    if takes_vocab:
        def build(config, vocab):
            return registry.get(config.ARCHITECTURE)(config, vocab)
    else:
        def build(config):
            return registry.get(config.ARCHITECTURE)(config)
    build.__doc__ = "Instantiate the class registered in {} under config.ARCHITECTURE.".format(registry.name)
    return build


# registry display names follow the reference (model_builder.py:4, encoder_builder.py:3, ...)
META_ARCHITECTURE = Registry("ARCHITECTURE")
META_ENCODER = Registry("ENCODER_LAYER")
META_DECODER = Registry("DECODER_LAYER")
META_ATTENTION = Registry("META_ATTENTION")
META_VISION_EMBEDDING = Registry("META_VISION_EMBEDDING")
META_TEXT_EMBEDDING = Registry("TEXT_EMBEDDING")

build_encoder = _factory(META_ENCODER, takes_vocab=False)
build_decoder = _factory(META_DECODER, takes_vocab=True)
build_attention = _factory(META_ATTENTION, takes_vocab=False)
build_vision_embedding = _factory(META_VISION_EMBEDDING, takes_vocab=False)
build_text_embedding = _factory(META_TEXT_EMBEDDING, takes_vocab=True)
_build_architecture = _factory(META_ARCHITECTURE, takes_vocab=True)


def build_model(config, vocab):
    """``config.ARCHITECTURE`` -> model on ``config.DEVICE`` (reference ``model_builder.py:6-10``); on
    PyTorch-ROCm the reference's ``DEVICE: cuda`` already names the HIP device."""
    return _build_architecture(config, vocab).to(torch.device(config.DEVICE))
