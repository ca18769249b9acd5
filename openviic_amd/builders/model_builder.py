"""``build_model(config, vocab)`` -- reference ``builders/model_builder.py:4-10``.

The model is constructed from ``config.ARCHITECTURE`` and moved to ``config.DEVICE``; on
PyTorch-ROCm the reference's ``DEVICE: cuda`` already names the HIP device.
"""
import torch

from .registry import Registry

META_ARCHITECTURE = Registry("ARCHITECTURE")


def build_model(config, vocab):
    model = META_ARCHITECTURE.get(config.ARCHITECTURE)(config, vocab)
    return model.to(torch.device(config.DEVICE))
