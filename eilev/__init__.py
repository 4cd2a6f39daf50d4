"""Import-path alias: lets the reference's callers (`from eilev.model.v2 import ...`, `from eilev.data.utils import ...`)
run unchanged against the MI355X-native implementation in `eilev_amd`.

This package only overrides the modules on the hot path (`eilev.model.v2`, `eilev.model.utils`, `eilev.data.utils`).  Every
other `eilev.*` module the user's scripts import (`eilev.data.frame`, `eilev.data.ego4d`, `eilev.model.v1`, ... — dataset
plumbing, out of scope here) falls through to whatever `eilev` distribution sits LATER on sys.path (the user's
checkout of the reference): `pkgutil.extend_path` appends those directories to this package's search path, so
`from eilev.data.frame import FrameInterleavedDataset` (ref:scripts/general/train_v2.py:21) resolves there.
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
