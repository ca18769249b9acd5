"""Position-wise feed-forward block -- reference ``models/modules/positionwise_feed_forward.py:5-28``."""
from torch import nn

from .. import ops


class PositionWiseFeedForward(nn.Module):
    """``LayerNorm(x + fc2(relu(fc1(x))))``; ``zero_rows`` additionally clears padded query rows
    (the ``masked_fill`` the reference applies right after this block, ``encoders.py:20`` /
    ``decoders.py:26``)."""

    def __init__(self, config):
        super().__init__()
        self.fc1 = nn.Linear(config.D_MODEL, config.D_FF)
        self.fc2 = nn.Linear(config.D_FF, config.D_MODEL)
        self.dropout = nn.Dropout(p=config.DROPOUT)
        self.dropout_2 = nn.Dropout(p=config.DROPOUT)
        self.layer_norm = nn.LayerNorm(config.D_MODEL)

    def forward(self, input, zero_rows=None):
        inner = ops.linear(input, self.fc1.weight, self.fc1.bias, relu=True)
        out = ops.linear(inner, self.fc2.weight, self.fc2.bias)
        return ops.layer_norm(out, self.layer_norm.weight, self.layer_norm.bias, residual=input,
                              zero_rows=zero_rows, eps=self.layer_norm.eps)
