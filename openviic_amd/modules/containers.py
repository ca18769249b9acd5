"""Stateful module containers (reference ``models/modules/containers.py:5-74``).

A *state* is a buffer with a default value.  ``statefulness(batch)`` expands every state to the
batch, runs the body and always restores the defaults (also when the body raises);
``apply_to_states(fn)`` rewrites every state in the tree, which is how beam search re-orders
per-beam state between steps.  The fused HIP beam search keeps its decode state in the engine
workspace instead and never touches these buffers; they exist for the step-wise API and so that
reference checkpoints (which contain the empty state buffers) load key-for-key.
"""
from contextlib import contextmanager
from typing import Callable, Iterator, Optional

import torch
from torch import nn


class Module(nn.Module):
    def __init__(self) -> None:
        super().__init__()
        self._is_stateful = False
        self._state_names = []
        self._state_defaults = {}

    def register_state(self, name: str, default: Optional[torch.Tensor]) -> None:
        self._state_names.append(name)
        self._state_defaults[name] = None if default is None else default.detach().clone()
        self.register_buffer(name, default)

    def _stateful_children(self) -> Iterator["Module"]:
        return (m for m in self.children() if isinstance(m, Module))

    def states(self) -> Iterator[Optional[torch.Tensor]]:
        for name in self._state_names:
            yield self._buffers[name]
        for child in self._stateful_children():
            yield from child.states()

    def apply_to_states(self, fn: Callable[[torch.Tensor], torch.Tensor]) -> None:
        for name in self._state_names:
            self._buffers[name] = fn(self._buffers[name])
        for child in self._stateful_children():
            child.apply_to_states(fn)

    def _default_on_device(self, name: str) -> Optional[torch.Tensor]:
        default = self._state_defaults[name]
        if default is None:
            return None
        current = self._buffers[name]
        device = current.device if current is not None else default.device
        return default.detach().clone().to(device)

    def enable_statefulness(self, batch_size: int) -> None:
        for child in self._stateful_children():
            child.enable_statefulness(batch_size)
        for name in self._state_names:
            value = self._default_on_device(name)
            if value is not None:
                value = value.unsqueeze(0).expand(batch_size, *value.shape).contiguous()
            self._buffers[name] = value
        self._is_stateful = True

    def disable_statefulness(self) -> None:
        for child in self._stateful_children():
            child.disable_statefulness()
        for name in self._state_names:
            self._buffers[name] = self._default_on_device(name)
        self._is_stateful = False

    @contextmanager
    def statefulness(self, batch_size: int):
        self.enable_statefulness(batch_size)
        try:
            yield
        finally:
            self.disable_statefulness()


class ModuleList(nn.ModuleList, Module):
    pass


class ModuleDict(nn.ModuleDict, Module):
    pass
