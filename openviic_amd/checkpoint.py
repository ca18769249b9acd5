"""Loading weights written by the reference's trainer.

``BaseTrainer.save_checkpoint`` (``trainers/base_trainer.py:138-153``) stores ``model.state_dict()``
under ``"state_dict"`` next to RNG / optimizer state.  Only the model weights are read here, with
``weights_only=True`` (nothing from the file is executed).  Key names and shapes are identical by
construction, so the load is ``strict=True`` by default -- including the empty decode-state
buffers the reference serialises.
"""
from typing import Union

import torch


def load_reference_checkpoint(model: torch.nn.Module, path_or_dict: Union[str, dict], strict: bool = True):
    ckpt = path_or_dict
    if isinstance(path_or_dict, str):
        ckpt = torch.load(path_or_dict, map_location="cpu", weights_only=True)
    state = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
    result = model.load_state_dict(state, strict=strict)
    if getattr(model, "_engine", None) is not None:
        model._engine = None            # pointer table is rebuilt on the next fused call
    return result
