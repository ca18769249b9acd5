"""Import path kept from the reference (``builders/text_embedding_builder.py``); defined in ``factories.py``."""
from .factories import META_TEXT_EMBEDDING, build_text_embedding  # noqa: F401
