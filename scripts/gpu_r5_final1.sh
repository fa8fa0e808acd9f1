#!/bin/bash
# round 5: the whole GPU suite + smoke + the default bench line on one box
mkdir -p gpurun_out/final1
timeout 3000 python -m pytest tests -q -m gpu -x > gpurun_out/final1/pytest.txt 2>&1; tail -5 gpurun_out/final1/pytest.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python bench.py 2>gpurun_out/final1/bench.err | tail -1 > gpurun_out/final1/bench.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/final1/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','blocking_ms_per_batch','two_in_flight_ms_per_step','rescored_per_query')}, d['roofline'].get('frac'), d['roofline'].get('traffic_note','')[:80])
print('c3', d['configs']['c3'])
print('c1', d['configs']['c1'])
PY
