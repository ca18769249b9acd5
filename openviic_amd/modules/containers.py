"""Stateful module containers (contract of the reference's ``models/modules/containers.py:5-74``).

A *state* is a buffer with a default value.  ``statefulness(batch)`` gives every state of the
module tree a leading batch dimension, runs the body and always restores the defaults (also when
the body raises); ``apply_to_states(fn)`` rewrites every state, which is how the host-loop beam
search re-orders per-beam state between steps.  The fused HIP beam search keeps its decode state
in the engine workspace instead and never touches these buffers; they exist for the step-wise API
and so that reference checkpoints (which contain the empty state buffers) load key-for-key.
"""
from contextlib import contextmanager
from typing import Callable, Dict, Iterator, Optional

import torch
from torch import nn


class Module(nn.Module):
    """``nn.Module`` with a table ``state name -> default tensor`` next to its buffers."""

    def __init__(self) -> None:
        super().__init__()
        self._is_stateful = False
        self._state_defaults: Dict[str, Optional[torch.Tensor]] = {}

    # -- declaration ----------------------------------------------------------------------------
    def register_state(self, name: str, default: Optional[torch.Tensor]) -> None:
        self._state_defaults[name] = default.detach().clone() if default is not None else None
        self.register_buffer(name, default)

    @property
    def _state_names(self):
        return list(self._state_defaults)

    # -- traversal ------------------------------------------------------------------------------
    def _state_owners(self) -> Iterator["Module"]:
        """This module, then its stateful children depth-first (only direct ``Module`` children
        recurse, as in the reference: a plain ``nn.Module`` in between hides its subtree)."""
        yield self
        for child in self.children():
            if isinstance(child, Module):
                yield from child._state_owners()

    def states(self) -> Iterator[Optional[torch.Tensor]]:
        for owner in self._state_owners():
            for name in owner._state_defaults:
                yield owner._buffers[name]

    def apply_to_states(self, fn: Callable[[torch.Tensor], torch.Tensor]) -> None:
        for owner in self._state_owners():
            for name in owner._state_defaults:
                owner._buffers[name] = fn(owner._buffers[name])

    # -- lifetime -------------------------------------------------------------------------------
    def _fresh(self, name: str, batch_size: Optional[int]) -> Optional[torch.Tensor]:
        default = self._state_defaults[name]
        if default is None:
            return None
        current = self._buffers.get(name)
        value = default.detach().clone().to(current.device if current is not None else default.device)
        if batch_size is not None:
            value = value.unsqueeze(0).expand(batch_size, *value.shape).contiguous()
        return value

    def _set_statefulness(self, batch_size: Optional[int]) -> None:
        for owner in self._state_owners():
            for name in owner._state_defaults:
                owner._buffers[name] = owner._fresh(name, batch_size)
            owner._is_stateful = batch_size is not None

    def enable_statefulness(self, batch_size: int) -> None:
        self._set_statefulness(int(batch_size))

    def disable_statefulness(self) -> None:
        self._set_statefulness(None)

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
