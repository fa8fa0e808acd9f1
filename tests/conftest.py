"""pytest configuration: markers + shared fixtures.

`-m "not gpu"` : oracle vs the reference's known-answer tests / golden vectors, host logic,
                 C-ABI symbol checks, world_size-2 gloo tests.  Runs without a GPU.
`-m gpu`       : parity tests proper — HIP path (through the C-ABI) vs the oracle.
"""
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as _o

    return _o.get()


@pytest.fixture(scope="session")
def oracle_portable():
    import oracle as _o

    return _o.get(portable=True)


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"
