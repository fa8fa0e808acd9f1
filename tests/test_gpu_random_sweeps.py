"""GPU: short runs of the randomised parity sweeps (`scripts/stress_parity.py`, `scripts/stress_ivf.py`) with seeds that
differ from the documented long runs: random shapes / metrics / modes through the C-ABI against the oracle, 0 mismatches."""
import re
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("script,seconds,seed", [("stress_parity.py", 20, 101), ("stress_ivf.py", 15, 202)])
def test_random_sweep_has_no_mismatch(script, seconds, seed):
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / script), str(seconds), str(seed)], capture_output=True,
                       text=True, timeout=600, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    m = re.search(r"cases (\d+).*mismatches (\d+)", r.stdout)
    assert m, r.stdout[-2000:]
    assert int(m.group(1)) > 20, r.stdout  # the sweep really ran
    assert int(m.group(2)) == 0, r.stdout[-4000:]
