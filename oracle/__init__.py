"""CPU oracle (test infrastructure).  See ``oracle/captioner.py``."""
from .captioner import OracleCaptioner  # noqa: F401
