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


@pytest.fixture(scope="session", autouse=True)
def _native_libraries():
    """make sure libepgx.so exists (a fresh checkout has no built artefacts: they are git-ignored);
    building needs hipcc, which both the build container and the GPU box have"""
    from epgpy_amd import _build

    if _build.needs_build():
        try:
            _build.build()
        except Exception as exc:  # pragma: no cover - reported by the tests that need the library
            print(f"could not build libepgx.so: {exc}")
    yield
