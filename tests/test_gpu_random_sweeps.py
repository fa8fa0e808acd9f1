"""GPU: short runs of the randomised parity sweeps (`scripts/stress_parity.py`, `scripts/stress_ivf.py`, `scripts/stress_inflight.py`) with seeds that
differ from the documented long runs: random shapes / metrics / modes through the C-ABI against the oracle, 0 mismatches."""
import re
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("script,cases,seed", [("stress_parity.py", 60, 101), ("stress_ivf.py", 60, 202), ("stress_inflight.py", 30, 303)])
def test_random_sweep_has_no_mismatch(script, cases, seed):
    # case-count-bounded ("c<N>"), not time-bounded: the case list depends on the seed only, not on the machine's speed
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / script), "c%d" % cases, str(seed)], capture_output=True,
                       text=True, timeout=900, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    m = re.search(r"cases (\d+).*mismatches (\d+)", r.stdout)
    assert m, r.stdout[-2000:]
    assert int(m.group(1)) == cases, r.stdout  # the sweep really ran
    assert int(m.group(2)) == 0, r.stdout[-4000:]
