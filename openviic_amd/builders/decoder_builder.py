"""Import path kept from the reference (``builders/decoder_builder.py``); defined in ``factories.py``."""
from .factories import META_DECODER, build_decoder  # noqa: F401
