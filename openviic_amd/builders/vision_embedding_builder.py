"""Import path kept from the reference (``builders/vision_embedding_builder.py``); defined in ``factories.py``."""
from .factories import META_VISION_EMBEDDING, build_vision_embedding  # noqa: F401
