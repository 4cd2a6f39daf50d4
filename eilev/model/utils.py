"""Alias of eilev_amd.model.utils under the reference's module path (ref:eilev/model/utils.py)."""
from eilev_amd.model.utils import process  # noqa: F401
