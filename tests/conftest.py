import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))

    return load


def pytest_sessionstart(session):
    """make sure libepgx.so exists and matches its sources (a fresh checkout has no built artefacts: they are git-ignored);
    building needs hipcc, which both the build container and the GPU box have.  Done HERE, not in a fixture: the build
    takes minutes and must not run under the time limit of whichever test happens to come first"""
    from epgpy_amd import _build

    if _build.needs_build():
        try:
            print("building libepgx.so ...", flush=True)
            _build.build()
        except Exception as exc:  # pragma: no cover - reported by the tests that need the library
            print(f"could not build libepgx.so: {exc}")
