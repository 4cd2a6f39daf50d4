from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)  # un-overridden submodules (frame, ego4d, ...) resolve from the user's own `eilev`
