from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)  # un-overridden submodules (v1, ...) resolve from the user's own `eilev`
