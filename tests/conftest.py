import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need an AMD GPU AND the built HIP library: skip them (instead of erroring) where either is missing, so a
    plain `pytest tests` works on a CPU-only machine.  On a GPU box a missing library is NOT skipped: it must fail loudly."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs an AMD GPU (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture()
def probes():
    """The probe build of the HIP library (alternative kernels behind `eilev_debug_*` switches, for A/B tests); while the fixture is
    active it also stands in for the product library inside eilev_amd (engines built in the test run on it), restored afterwards."""
    from eilev_amd import abi

    lib = abi.load_probes()
    if lib is None:
        pytest.skip("libeilev_hip_probes.so not built (python eilev_amd/csrc/build.py --variant probes -DEILEV_PROBES)")
    saved = abi._hip
    abi._hip = lib
    try:
        yield lib
    finally:
        abi._hip = saved
