"""ThreadSanitizer over the HOST state machine of the library ("race detection / sanitizers — the build must add its own").

The reference leans on Rust's `Send + Sync` (src/index/mod.rs:78, src/python/mod.rs:950); the C++ host half of csrc/lynse_hip.hip — reader /
writer locks, search-context leasing, tickets, the IVF index guard, lazy builds, bounded waits — gets `hipcc --cuda-host-only
-fsanitize=thread` against a host-only stand-in for the HIP runtime and for RCCL (tests/hipstub/: TEST INFRASTRUCTURE, no GPU, kernels
do not run) and a driver that runs the concurrent-reader / in-flight / insert-versus-ticket scenarios of the GPU suite from several
threads, plus the dead-peer scenario: with a 2-rank communicator whose peer never arrives every wait that ends in a collective — FLAT and
IVF tickets, the blocking sharded searches, the all-reduce of the sharded k-means, the communicator's self-check — must come back with
LYNSE_ERR_TIMEOUT within the configured bound (src/cluster.rs:243-261: the reference's coordinator retries once and errors).

Findings of the first runs, all fixed (regressions show up here as ThreadSanitizer reports): unsynchronised once-flags of the launch
helpers; lynse_hip_ivf_set_row_map / assign / profile getters outside the index guard; `kk = min(k, h->n)` formed before the reader lock
in submit; the profiling flags; a lock-order inversion between the communicator's mutex and the shard's lock (submit against the blocking
sharded search)."""
import os
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "build" / "tsan"


def _build():
    r = subprocess.run(["make", "-C", str(ROOT / "tests" / "hipstub")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])


@pytest.fixture(scope="module")
def driver():
    _build()
    exe = OUT / "tsan_driver"
    assert exe.exists()
    return exe


@pytest.mark.parametrize("scenario", ["readers", "tickets", "ivf", "comm1", "dead_peer"])
def test_host_state_machine_under_threadsanitizer(driver, scenario):
    env = dict(os.environ)
    env["LYNSE_HIP_RCCL_PATH"] = str(OUT / "librcclstub.so")
    env["TSAN_OPTIONS"] = "halt_on_error=0 second_deadlock_stack=1"
    p = subprocess.run([str(driver), scenario], env=env, capture_output=True, text=True, timeout=600)
    assert "ThreadSanitizer" not in p.stderr, p.stderr[-6000:]
    assert "CHECK FAILED" not in p.stderr, p.stderr[-3000:]
    assert p.returncode == 0, p.stderr[-3000:]
    assert "[tsan_driver] %s: 0 failed checks" % scenario in p.stderr
