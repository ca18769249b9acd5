"""Import path kept from the reference (``builders/attention_builder.py``); defined in ``factories.py``."""
from .factories import META_ATTENTION, build_attention  # noqa: F401
