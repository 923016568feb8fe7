"""MI355X-native (gfx950) implementation of the OnePose++ 2D-3D matching forward.

Public surface mirrors the reference package `src.models.OnePosePlus`
(/root/reference/src/models/OnePosePlus/__init__.py:1): `OnePosePlus_model`.
"""
from .config import default_config  # noqa: F401
from .model import OnePosePlus_model  # noqa: F401

__all__ = ["OnePosePlus_model", "default_config"]
