"""Input containers of the model API.

``Instance`` / ``InstanceList`` carry the batched tensors a model reads by attribute
(``region_features``, ``region_boxes``, ``grid_features``, ``caption_tokens``), as in the reference
``utils/instance.py:9-30,32-178``.  Collating a list of per-image ``Instance`` objects zero-pads
ragged first dimensions (``utils/instance.py:156-171``); the zero rows are what the vision
embedding later turns into the padding mask.
"""
from collections import OrderedDict
from typing import Any, List, Sequence

import numpy as np
import torch


class Instance(OrderedDict):
    """One sample: an ordered attribute dictionary."""

    def __init__(self, **fields: Any):
        super().__init__(fields)

    def __getattr__(self, key: str) -> Any:
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key) from None

    def __setattr__(self, key: str, value: Any) -> None:
        self[key] = value

    def get_fields(self) -> List[str]:
        return list(self.keys())


def _pad_rows(values: Sequence[torch.Tensor], padding_value: float = 0) -> torch.Tensor:
    """Stack along a new leading dimension, zero-padding ragged first dimensions.  The filler is float32, as in the
    reference (``utils/instance.py:156-171``: ``torch.zeros(...)``), so ``torch.cat``'s type promotion applies: a padded
    float64 field stays float64, a padded integer field becomes float32, an unpadded one keeps its dtype."""
    longest = max(v.shape[0] for v in values)
    out = []
    for v in values:
        missing = longest - v.shape[0]
        if missing:
            filler = torch.full((missing,) + tuple(v.shape[1:]), padding_value, dtype=torch.float32)
            v = torch.cat([v, filler], dim=0)
        out.append(v.unsqueeze(0))
    return torch.cat(out, dim=0)


class InstanceList(OrderedDict):
    """A batch: every tensor field is stacked along a new leading dimension."""

    def __init__(self, instance_list: Sequence[Instance] = ()):
        super().__init__()
        if len(instance_list) == 0:
            return
        assert all(isinstance(i, Instance) for i in instance_list)
        for key in instance_list[0].get_fields():
            values = [inst[key] for inst in instance_list]
            first = values[0]
            if isinstance(first, np.ndarray):
                values = _pad_rows([torch.as_tensor(v) for v in values])
            elif isinstance(first, torch.Tensor):
                values = _pad_rows(values)
            self[key] = values

    def __getattr__(self, name: str) -> Any:
        if name.startswith("_") or name not in self:
            return None
        return self[name]

    def __setattr__(self, name: str, value: Any) -> None:
        if name.startswith("_"):
            super().__setattr__(name, value)
        else:
            self[name] = value

    def set(self, name: str, value: Any) -> None:
        self[name] = value

    def has(self, name: str) -> bool:
        return name in self

    def remove(self, name: str) -> None:
        del self[name]

    def get_fields(self) -> List[str]:
        return list(self.keys())

    @property
    def batch_size(self) -> int:
        for value in self.values():
            if isinstance(value, torch.Tensor):
                return value.shape[0]
            if isinstance(value, list):
                return len(value)
        return 0

    def _map(self, method: str, *args: Any, **kwargs: Any) -> "InstanceList":
        out = InstanceList()
        for key, value in self.items():
            out[key] = getattr(value, method)(*args, **kwargs) if hasattr(value, method) else value
        return out

    def to(self, *args: Any, **kwargs: Any) -> "InstanceList":
        return self._map("to", *args, **kwargs)

    def unsqueeze(self, *args: Any, **kwargs: Any) -> "InstanceList":
        return self._map("unsqueeze", *args, **kwargs)

    def squeeze(self, *args: Any, **kwargs: Any) -> "InstanceList":
        return self._map("squeeze", *args, **kwargs)

    def __repr__(self) -> str:
        return "InstanceList(fields=[{}])".format(", ".join(self.keys()))
