"""Import path kept from the reference (``builders/encoder_builder.py``); defined in ``factories.py``."""
from .factories import META_ENCODER, build_encoder  # noqa: F401
