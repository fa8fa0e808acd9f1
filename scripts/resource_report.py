#!/usr/bin/env python3
"""Print registers / scratch / occupancy of the kernels matching a substring from csrc/resource_usage.txt."""
import re
import subprocess
import sys
from pathlib import Path

pat = sys.argv[1] if len(sys.argv) > 1 else "k_scan_h16"
txt = (Path(__file__).resolve().parent.parent / "lynsedb_amd/csrc/resource_usage.txt").read_text()
for b in re.split(r"(?=remark: [^\n]*Function Name:)", txt):
    m = re.search(r"Function Name: (\S+)", b)
    if not m or pat not in m.group(1):
        continue
    dem = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    dem = dem.replace("void lynse::", "").replace("(lynse::ScanArgs)", "")

    def g(k):
        mm = re.search(k + r": (\d+)", b)
        return int(mm.group(1)) if mm else None

    scratch, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
    print(f"{dem[:100]:100s} VGPR {g('VGPRs')} AGPR {g('AGPRs')} scratch {scratch} SGPR {g('SGPRs')} occ {occ} LDS {lds}")
