"""pytest config: registers the ``gpu`` marker and puts the repo root on sys.path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "soak: randomised many-seed runs (also marked gpu; ICEM_SOAK_SEEDS scales them)")


@pytest.fixture(autouse=True)
def _library_options_back_to_defaults():
    """The library's development options (icem_set_option) are process-wide: every test starts and ends on the defaults.
    Tests flip them with ``icem_amd._lib.set_option`` -- the library itself reads no environment variable."""
    from icem_amd import _lib as L
    built = os.path.exists(L.lib_path())
    if built:
        L.reset_options()
    yield
    if built:
        L.reset_options()


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
