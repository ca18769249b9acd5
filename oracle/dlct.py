"""CPU oracle for the dual-collaborative (DLCT) embedding + encoder  --  TEST INFRASTRUCTURE, NOT THE PRODUCT.

Restates ``GeometricDualFeatureEmbedding`` (models/modules/vision_embeddings.py:46-71),
``get_combine_masks`` / ``get_grids_by_corner`` (models/utils.py:100-154) and
``DualCollaborativeLevelEncoder.forward`` (models/modules/encoders.py:153-211) on a reference-format
``state_dict``.  Neither class runs in the reference as shipped; this file applies
the same three repairs as the fixture generator (``tests/golden/make_goldens.py::g8_dlct_encoder``):

  1. the region->grid visibility mask is (B,1,n,g*g), not (B,1,1,n,g*g);
  2. key-padding masks (B,1,1,n) are expanded over the query dimension before concatenation with it;
  3. after a cross layer the rows cleared are those of the *query* side's padding mask.

Parity pinning: ``tests/golden/g8_dlct_encoder*.npz`` (outputs of the reference's own sub-modules composed with
these repairs); checked by ``tests/test_oracle_golden.py``.  Only ``tests/`` may import this file.
"""
from __future__ import annotations

import math
from typing import Any, Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .captioner import OracleCaptioner, _get, box_relation_features, padding_mask_from_features

Tensor = torch.Tensor


def _cell_of(edges: Tensor, x: Tensor) -> int:
    """models/utils.py:100-111 -- index of the last cell edge that is <= x (0 when there is none).  The edges are
    float64 ``arange(g)/g`` but the comparison against a float32 tensor element happens in float32."""
    cell = 0
    for i in range(edges.numel()):
        if bool(edges[i] <= x):
            cell = i
    return cell


def grid_visibility_mask(boxes: Tensor, grid_size: int) -> Tensor:
    """models/utils.py:113-154 -- True where a grid cell is NOT covered by the region box.  (B,1,n,g*g); repair 1."""
    edges = torch.from_numpy(np.arange(grid_size) / grid_size).to(torch.float32)
    bsz, n, _ = boxes.shape
    mask = torch.ones(bsz, 1, n, grid_size * grid_size, dtype=torch.bool)
    for b in range(bsz):
        for i in range(n):
            x_min, y_min, x_max, y_max = boxes[b, i]
            col0, row0 = _cell_of(edges, x_min), _cell_of(edges, y_min)
            col1, row1 = _cell_of(edges, x_max), _cell_of(edges, y_max)
            for row in range(row0, row1 + 1):
                for col in range(col0, col1 + 1):
                    mask[b, 0, i, row * grid_size + col] = False
    return mask


def normalized_position_encoding(batch: int, n: int, d_model: int, temperature: float = 10000.0,
                                 scale: float = 2 * math.pi, dtype: torch.dtype = torch.float32) -> Tensor:
    """models/modules/pos_embeddings.py:58-72 with mask=None, normalize=True: position i+1 is divided by
    (n + 1e-6) and multiplied by ``scale`` before the sinusoids."""
    position = torch.ones(batch, n).cumsum(1, dtype=dtype)
    position = position / (position[:, -1:] + 1e-6) * scale
    channel = torch.arange(d_model, dtype=dtype)
    divisor = temperature ** (2 * torch.div(channel, 2, rounding_mode="floor") / d_model)
    angle = position[:, :, None] / divisor
    return torch.stack((angle[:, :, 0::2].sin(), angle[:, :, 1::2].cos()), dim=-1).flatten(-2)


class OracleDualEncoder(OracleCaptioner):
    """``embedding_sd`` / ``encoder_sd`` are the state_dicts of the two reference modules."""

    def __init__(self, encoder_cfg: Any, embedding_sd: Dict[str, Tensor], encoder_sd: Dict[str, Tensor],
                 dtype: torch.dtype = torch.float32):
        # dtype = float64: the conditioning yardstick of OracleCaptioner (masks are still found on the fp32 inputs)
        self.dtype = dtype
        self.sd = {"emb." + k: v.detach().to(dtype).cpu() for k, v in embedding_sd.items()}
        self.sd.update({"enc." + k: v.detach().to(dtype).cpu() for k, v in encoder_sd.items()})
        self.cfg = encoder_cfg
        self.trace = None
        self.d_model = int(_get(encoder_cfg, "D_MODEL"))
        self.heads = int(_get(encoder_cfg, "HEAD"))
        self.layers = int(_get(encoder_cfg, "LAYERS"))
        self.trig = bool(_get(encoder_cfg, "TRIGNOMETRIC_EMBEDDING"))
        self.self_att = _get(encoder_cfg, "SELF_ATTENTION")
        self.cross_att = _get(encoder_cfg, "CROSS_ATTENTION")

    def embed(self, region: Tensor, region_boxes: Tensor, grid: Tensor, grid_boxes: Tensor):
        """vision_embeddings.py:56-71 (+ repairs 1, 2)."""
        bsz, n = region.shape[:2]
        gg = grid.shape[1]
        region_mask, grid_mask = padding_mask_from_features(region), padding_mask_from_features(grid)
        r2g = grid_visibility_mask(region_boxes, int(gg ** 0.5))
        region2all = torch.cat([region_mask.expand(bsz, 1, n, n), r2g], dim=-1)
        grid2all = torch.cat([r2g.permute(0, 1, 3, 2), grid_mask.expand(bsz, 1, gg, gg)], dim=-1)
        return ((self._lin("emb.region_proj", region.to(self.dtype)), region_mask),
                (self._lin("emb.grid_proj", grid.to(self.dtype)), grid_mask), (region2all, grid2all))

    def geometry(self, boxes: Tensor) -> Tensor:
        """encoders.py:157-164."""
        d_g = self.d_model // self.heads if self.trig else 4
        emb = box_relation_features(boxes.to(self.dtype), dim_g=d_g, trignometric=self.trig)
        b, n = emb.shape[:2]
        per_head = [self._lin("enc.fc_gs.%d" % i, emb.view(-1, d_g)).view(b, 1, n, n) for i in range(self.heads)]
        return F.relu(torch.cat(per_head, dim=1))

    def encode(self, region: Tensor, region_boxes: Tensor, region_mask: Tensor, region2all: Tensor,
               grid: Tensor, grid_boxes: Tensor, grid_mask: Tensor, grid2all: Tensor) -> Tuple[Tensor, Tensor]:
        """encoders.py:153-211 (+ repair 3)."""
        n = region.shape[1]
        w = self.geometry(torch.cat([region_boxes, grid_boxes], dim=1))
        self._note("geometry_weights", w)

        def pe(x):
            return normalized_position_encoding(x.shape[0], x.shape[1], self.d_model, dtype=self.dtype)

        def layer(prefix, cfg, q, kv, geo, mask, q_pad):
            att = self.multi_head(prefix + ".mhatt", cfg, q, kv, kv, mask, geo)
            return self.feed_forward(prefix + ".pwff", att).masked_fill(q_pad[:, 0, 0, :, None], 0)
        rf = self._ln("enc.layer_norm_region", region) + pe(region)
        gf = self._ln("enc.layer_norm_grid", grid) + pe(grid)
        for i in range(self.layers):
            rf = layer("enc.layers_region.%d" % i, self.self_att, rf, rf, w[:, :, :n, :n], region_mask, region_mask)
            gf = layer("enc.layers_grid.%d" % i, self.self_att, gf, gf, w[:, :, n:, n:], grid_mask, grid_mask)
            both = torch.cat([rf, gf], dim=1)
            both = both + pe(both)
            rf = layer("enc.region2grid.%d" % i, self.cross_att, rf, both, w[:, :, :n, :], region2all, region_mask)
            gf = layer("enc.grid2region.%d" % i, self.cross_att, gf, both, w[:, :, n:, :], grid2all, grid_mask)
            self._note("layer%d_region" % i, rf)
            self._note("layer%d_grid" % i, gf)
        return torch.cat([rf, gf], dim=1), torch.cat([region_mask, grid_mask], dim=-1)
