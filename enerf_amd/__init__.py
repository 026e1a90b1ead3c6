"""enerf_amd — MI355X-native (gfx950) ENeRF rendering hot path behind the reference's Python surface."""
from .config import CascadeConfig, EnerfConfig  # noqa: F401

__all__ = ["CascadeConfig", "EnerfConfig"]
