#!/usr/bin/env python3
"""Runs ONE of bench.py's extra configurations (c1 / c3 / c5_share / c4_share) and prints its JSON: python scripts/other_config.py c4_share"""
import json, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["LYNSE_BENCH_ONLY_CONFIG"] = sys.argv[1]
import torch  # noqa: E402
import bench  # noqa: E402
print(json.dumps(bench.other_configs(torch.device("cuda", 0))))
