"""`bench.py --config c4 / c5` (and the headline c2) end to end at reduced sizes (one GPU, and 2 / 4 / 8 gloo ranks sharing it — the
first real 8-GPU run must not also be the first run of the 8-rank code path): the line is printed, names the
workload BASELINE.json names, and its own verification legs are green (tickets == the blocking search; recall against the exact
top-k of the whole collection for IVF, the oracle's packed search for Hamming)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def run_bench(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    p = subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, cwd=str(ROOT), env=e, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("gpus", [1, 2, 4])
def test_bench_config_c4_prints_a_verified_line(gpus):
    env = {"LYNSE_BENCH_BACKEND": "gloo"} if gpus > 1 else None       # two ranks share the one GPU of a test box: no RCCL between them
    d = run_bench(["--config", "c4", "--gpus", str(gpus), "--rows-per-gpu", "150000", "--nlist", "512", "--nprobe", "16", "--steps", "4", "--warmup", "1"], env)
    assert d["n_gpus"] == gpus and d["scaling"] == "weak" and d["unit"] == "queries/s" and d["value"] > 0
    assert "IVF-Flat IP %dx768" % (150000 * gpus) in d["metric"] and d["config"]["lists_trained"] == 512
    assert d["verify"]["tickets_equal_blocking_search"] is True
    assert d["verify"]["recall_at_k_vs_exact_top_k_of_the_collection"] >= 0.8      # (IVF is approximate: 512 lists over 4096 generating centres, 2 Lloyd iterations)
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["achieved"] > 0
    if gpus == 1:
        assert d["tickets"]["in_flight"] > 0 and d["tickets"]["redone_in_wait"] == 0
        assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["value"] > 0, d["cpu_baseline"]


@pytest.mark.parametrize("gpus", [1, 2, 8])
def test_bench_config_c5_prints_a_verified_line(gpus):
    env = {"LYNSE_BENCH_BACKEND": "gloo"} if gpus > 1 else None
    d = run_bench(["--config", "c5", "--gpus", str(gpus), "--rows", str(600000 * gpus), "--steps", "4", "--warmup", "1", "--no-cpu-baseline"], env)
    assert d["n_gpus"] == gpus and d["dtype"] == "u64" and d["value"] > 0
    assert "Hamming %dx1024-bit" % (600000 * gpus) in d["metric"] and "k=50" in d["metric"]
    assert d["verify"] == {"tickets_equal_blocking_search": True, "oracle_bit_exact_on_200k_row_sample": True}


@pytest.mark.parametrize("gpus", [4, 8])
def test_bench_headline_runs_sharded_over_4_and_8_ranks(gpus):
    """The headline workload (FLAT-IP, 256 queries, k = 10) row-sharded over 4 / 8 gloo ranks that share the one GPU: tickets in flight, the
    exchange (torch all-gather under gloo) + canonical merge, recall against the exact top-k of the WHOLE collection."""
    d = run_bench(["--gpus", str(gpus), "--rows", "400000", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-configs", "--settle-ms", "0"],
                  {"LYNSE_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == gpus and d["scaling"] == "strong" and d["value"] > 0
    assert d["verify"]["recall_at_k_tolerant"] == 1.0, d["verify"]
