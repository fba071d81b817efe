import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Every native call of the GPU suite starts from a workspace filled with 0xFF bytes (every float a NaN, every statistics limb -1):
# a kernel that reads scratch the call did not write shows up as NaN / MiddError instead of depending on what an earlier call
# with another layout left behind (modules.py reads the variable when a model is constructed).
os.environ.setdefault("MIDD_POISON_WS", "255")

import midd_loader  # noqa: E402

midd_loader.load()

REFERENCE_DIR = "/root/reference/Backend"
HAVE_REFERENCE = os.path.isdir(REFERENCE_DIR)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    skip_ref = pytest.mark.skip(reason="/root/reference not present (GPU box)")
    for item in items:
        if "reference" in item.keywords and not HAVE_REFERENCE:
            item.add_marker(skip_ref)


@pytest.fixture(scope="session")
def reference_module():
    """Imports the reference's DDIM module in-process (SURVEY.md App. B recipe)."""
    if not HAVE_REFERENCE:
        pytest.skip("/root/reference not present")
    from tests.golden.ref_import import import_reference
    return import_reference()
