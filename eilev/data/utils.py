"""Alias of eilev_amd.data.utils under the reference's module path (ref:eilev/data/utils.py)."""
from eilev_amd.data.utils import (  # noqa: F401
    clean_narration_text,
    generate_input_ids_and_labels,
    generate_input_ids_and_labels_from_interleaved,
)


def __getattr__(name):
    import eilev_amd.data.utils as _u

    return getattr(_u, name)
