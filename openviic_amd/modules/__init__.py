"""Host-side mirror of the reference's ``models.modules`` (registered under the same names)."""
from .containers import Module, ModuleList, ModuleDict
from .attentions import (ScaledDotProductAttention, AugmentedGeometryScaledDotProductAttention,
                         AugmentedMemoryScaledDotProductAttention, MultiHeadAttention)
from .feed_forward import PositionWiseFeedForward
from .embeddings import (FeatureEmbedding, DualFeatureEmbedding, GeometricDualFeatureEmbedding, UsualEmbedding,
                         SinusoidPositionalEmbedding, grid_visibility_mask)
from .encoders import (EncoderLayer, Encoder, MultilevelEncoder, GeometricEncoder,
                       DualCollaborativeLevelEncoder)
from .decoders import DecoderLayer, MeshedDecoderLayer, Decoder, MeshedDecoder, sinusoid_encoding_table
