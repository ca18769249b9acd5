"""openviic_amd -- MI355X-native (gfx950) captioning engine with OpenViIC's model API.

    from openviic_amd.builders import build_model
    from openviic_amd.config import get_config

Importing the package registers every module / architecture class under the reference's names.
"""
from . import builders  # noqa: F401  (populates the registries)
from .config import ConfigNode, get_config, model_config  # noqa: F401
from .instance import Instance, InstanceList  # noqa: F401

__version__ = "0.1.0"
