"""Import path kept from the reference (``builders/model_builder.py``); defined in ``factories.py``."""
from .factories import META_ARCHITECTURE, build_model  # noqa: F401
