"""Deterministic synthetic weights, vocabulary stub and inputs for tests and the benchmark.

There is no network for checkpoints or datasets, so the benchmark and the parity tests run on
random-initialised weights of the reference architectures.  The values must be identical in
three places -- the imported reference (golden generation), the CPU oracle and the HIP engine --
so they are drawn per tensor from a numpy ``Generator`` seeded by ``crc32(name) ^ seed`` and
loaded everywhere through ``load_state_dict``.

``mode="reference_init"`` reproduces the distributions the reference constructors produce
(SURVEY.md section 8d): Xavier-uniform attention / gate weights with zero bias
(``models/modules/attentions.py:34-42``, ``encoders.py:86-91``, ``decoders.py:46-49``), memory
slots N(0, 1/d_k) and N(0, 1/m) (``attentions.py:151-152``), PyTorch-default U(+-1/sqrt(fan_in))
for every other Linear, N(0,1) embeddings with a zero padding row (``text_embeddings.py:15``),
LayerNorm (1, 0).  ``mode="generic"`` additionally randomises every bias and LayerNorm affine so
that a kernel which drops one of them cannot pass a parity test.
"""
from __future__ import annotations

import math
import re
import zlib
from typing import Dict, Mapping, Optional, Tuple

import numpy as np
import torch


class SyntheticVocab:
    """The five attributes the model reads from a vocabulary (``data_utils/vocab.py:41,61-64``)."""

    padding_idx = 0
    bos_idx = 1
    eos_idx = 2
    unk_idx = 3

    def __init__(self, size: int = 10201, max_caption_length: int = 20):
        self._size = int(size)
        self.max_caption_length = int(max_caption_length)

    def __len__(self) -> int:
        return self._size


_SKIP = re.compile(r"(running_keys|running_values|running_mask_self_attention|running_seq|pos_emb\.weight)$")


def _rng(name: str, seed: int) -> np.random.Generator:
    return np.random.default_rng([zlib.crc32(name.encode()), seed])


def _uniform(rng, shape, bound) -> np.ndarray:
    return ((rng.random(shape, dtype=np.float64) * 2.0 - 1.0) * bound).astype(np.float32)


def synthetic_tensor(name: str, shape: Tuple[int, ...], seed: int, mode: str,
                     memory_dims: Optional[Tuple[int, int]] = None) -> Optional[np.ndarray]:
    """Value of parameter ``name``; ``None`` for buffers that are not parameters."""
    if _SKIP.search(name):
        return None
    rng = _rng(name, seed)
    generic = mode == "generic"
    leaf = name.rsplit(".", 1)[-1]
    if "layer_norm" in name:
        if leaf == "weight":
            return (1.0 + _uniform(rng, shape, 0.5)) if generic else np.ones(shape, np.float32)
        return _uniform(rng, shape, 0.2) if generic else np.zeros(shape, np.float32)
    if leaf == "m_k" or leaf == "m_v":
        d_k, m = memory_dims if memory_dims else (64, shape[1])
        std = 1.0 / d_k if leaf == "m_k" else 1.0 / m
        return (rng.standard_normal(shape) * std).astype(np.float32)
    if name.endswith("word_emb.components.weight"):
        table = rng.standard_normal(shape).astype(np.float32)
        table[0] = 0.0
        return table
    xavier = bool(re.search(r"attention\.fc_[qkvo]\.|fc_gs\.|fc_alphas\.", name))
    if leaf == "weight":
        fan_out, fan_in = shape[0], int(np.prod(shape[1:]))
        bound = math.sqrt(6.0 / (fan_in + fan_out)) if xavier else 1.0 / math.sqrt(fan_in)
        return _uniform(rng, shape, bound)
    if leaf == "bias":
        if xavier and not generic:
            return np.zeros(shape, np.float32)
        return _uniform(rng, shape, 0.05 if xavier else 0.04)
    raise KeyError("no synthetic rule for parameter '{}'".format(name))


def synthetic_state_dict(template: Mapping[str, torch.Tensor], seed: int = 1234,
                         mode: str = "reference_init",
                         memory_dims: Optional[Tuple[int, int]] = None) -> Dict[str, torch.Tensor]:
    """Deterministic values for every parameter named in ``template`` (a ``state_dict``)."""
    out: Dict[str, torch.Tensor] = {}
    for name, tensor in template.items():
        value = synthetic_tensor(name, tuple(tensor.shape), seed, mode, memory_dims)
        if value is not None:
            out[name] = torch.from_numpy(np.ascontiguousarray(value))
    return out


def eos_biased_state_dict(state_dict: Mapping[str, torch.Tensor], template: Mapping[str, torch.Tensor], eos_idx: int = 2,
                          ramp: float = 2.0, gain: float = 3.0, mid: int = 10, seed: int = 5) -> Dict[str, torch.Tensor]:
    """Synthetic weights whose captions END at realistic lengths.  Random-init weights never emit ``<eos>``,
    so nothing about finished beams -- the -999 branch, early exit -- shows on them.  Here the decoder's position table (a
    ``state_dict`` entry: ``decoder.pos_emb.weight``, taken from ``template``, the model's own ``state_dict()``) gets a component
    ``ramp * (t - mid) * u`` along a fixed unit direction ``u``, and the ``<eos>`` row of the vocabulary projection is ``gain * u``:
    the ``<eos>`` logit rises with the position and the beams end around step ``mid`` (with the defaults and a 2-layer d = 128
    model: between steps 6 and 12, mean 9).  Returns a copy; works for the oracle and the product alike."""
    out = {k: v.clone() for k, v in state_dict.items()}
    pos = template["decoder.pos_emb.weight"].detach().clone().float().cpu()
    g = torch.Generator().manual_seed(seed)
    u = torch.randn(pos.shape[1], generator=g)
    u = u / u.norm()
    steps = torch.arange(pos.shape[0], dtype=torch.float32)[:, None]
    out["decoder.pos_emb.weight"] = pos + ramp * (steps - mid) * u
    fc = out["decoder.fc.weight"].clone()
    fc[eos_idx] = gain * u
    out["decoder.fc.weight"] = fc
    return out


def synthetic_features(batch: int, regions: int = 50, d_feature: int = 2048, seed: int = 0,
                       ragged: bool = False) -> torch.Tensor:
    """``randn(B, N, d)`` from a CPU generator; ``ragged`` zero-pads tails."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, regions, d_feature, generator=g)
    if ragged:
        for b in range(batch):
            keep = regions - (b * 3) % max(1, regions // 2)
            x[b, keep:] = 0
    return x


def synthetic_boxes(batch: int, regions: int = 50, seed: int = 0) -> torch.Tensor:
    """(x_min, y_min, x_max, y_max) in [0, 1]: xy = rand*0.5, wh = rand*0.5."""
    g = torch.Generator().manual_seed(seed + 1)
    xy = torch.rand(batch, regions, 2, generator=g) * 0.5
    wh = torch.rand(batch, regions, 2, generator=g) * 0.5
    return torch.cat([xy, xy + wh], dim=-1)


def synthetic_dual_inputs(batch: int, regions: int, grid: int, d_region: int, d_grid: int, seed: int = 0):
    """Inputs of the dual-collaborative encoder: ragged region features with boxes (padding = zero rows and zero
    boxes), full ``grid x grid`` features and the cell boxes ``(c/g, r/g, (c+1)/g, (r+1)/g)`` in row-major order.
    Two boxes of image 0 are fixed: one covering every cell, one with all corners exactly on cell edges."""
    region = synthetic_features(batch, regions, d_region, seed=seed, ragged=True)
    cells = synthetic_features(batch, grid * grid, d_grid, seed=seed + 1)
    boxes = synthetic_boxes(batch, regions, seed=seed)
    boxes[region.abs().sum(-1) == 0] = 0
    boxes[0, 0] = torch.tensor([0.0, 0.0, 0.999, 0.999])
    boxes[0, 1] = torch.tensor([1.0 / grid, 2.0 / grid, 1.0 / grid, 2.0 / grid])
    edge = torch.arange(grid, dtype=torch.float32) / grid
    ys, xs = torch.meshgrid(edge, edge, indexing="ij")
    cell_boxes = torch.stack([xs, ys, xs + 1.0 / grid, ys + 1.0 / grid], dim=-1).reshape(1, grid * grid, 4)
    return region, boxes, cells, cell_boxes.repeat(batch, 1, 1)
