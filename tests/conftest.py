import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("BSK_PY_WATCH_ENV", "1")  # the tests flip BSK_* switches inside one process: bio_amd.sketches.Engine re-reads them when they change
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def engine():
    """One GPU context for the whole session.  Fails loudly (no CPU fallback)."""
    from bio_amd import sketches as S
    return S.Engine(0)
