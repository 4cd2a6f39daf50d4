"""Alias of eilev_amd.model.v2 under the reference's module path (ref:eilev/model/v2.py)."""
from eilev_amd.model.v2 import VideoBlipForConditionalGeneration, VideoBlipVisionModel  # noqa: F401
