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


def oracle_for_every_query(fn, nq, workers=None):
    """[fn(0), ..., fn(nq - 1)] — the oracle's single-query calls of a whole batch, run from a thread pool: the calls are plain C through
    ctypes (the GIL is released), re-entrant, and memory-bound, so the host's cores answer all 256 queries of a BASELINE batch in the
    time the single-threaded loop needed for a dozen (the config tests checked 4-10 of 256 queries against the oracle)."""
    from concurrent.futures import ThreadPoolExecutor

    workers = workers or max(1, min(32, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    with ThreadPoolExecutor(max_workers=workers) as ex:
        return list(ex.map(fn, range(nq)))
