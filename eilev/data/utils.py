"""Alias of eilev_amd.data.utils under the reference's module path (ref:eilev/data/utils.py).

Names this repo does not implement (NarratedActionClipSampler: pytorchvideo clip sampling for the raw-video datasets, out of
scope) are looked up in the `eilev/data/utils.py` of the user's own `eilev` distribution further down sys.path, if any."""
from eilev_amd.data.utils import (  # noqa: F401
    clean_narration_text,
    generate_chunks,
    generate_input_ids_and_labels,
    generate_input_ids_and_labels_from_interleaved,
    parse_timestamp,
)

_fallback = None


def _user_module():
    """The next eilev/data/utils.py on the (extended) package path, loaded under a private name."""
    global _fallback
    if _fallback is None:
        import importlib.util
        import os

        import eilev.data as _pkg

        here = os.path.dirname(os.path.abspath(__file__))
        for d in list(_pkg.__path__):
            cand = os.path.join(d, "utils.py")
            if os.path.abspath(d) != here and os.path.exists(cand):
                spec = importlib.util.spec_from_file_location("eilev.data._user_utils", cand)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                _fallback = mod
                break
        else:
            _fallback = False
    return _fallback


def __getattr__(name):
    import eilev_amd.data.utils as _u

    try:
        return getattr(_u, name)
    except AttributeError:
        pass
    if name.startswith("__"):
        raise AttributeError(name)
    mod = _user_module()
    if mod and hasattr(mod, name):
        return getattr(mod, name)
    raise AttributeError(f"module 'eilev.data.utils' has no attribute {name!r} (not part of the MI355X-native hot path and no other "
                         "`eilev` distribution on sys.path provides it)")
