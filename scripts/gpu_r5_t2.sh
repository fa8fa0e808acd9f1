#!/bin/bash
# round 5: the whole GPU suite on the current binary + bench --config c4 (20 Lloyd rounds, second data set) + the c4_share entry of the default line
mkdir -p gpurun_out/t2
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/t2/pytest.txt 2>&1; grep -E "passed|failed|error" gpurun_out/t2/pytest.txt | tail -3
timeout 1200 python bench.py --config c4 --steps 30 --warmup 3 > gpurun_out/t2/bench_c4.json 2> gpurun_out/t2/bench_c4.err; tail -3 gpurun_out/t2/bench_c4.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/t2/bench_c4.json").read().strip().splitlines()[-1])
print("c4:", d["ms_per_step"], d["value"], d["config"]["train_s"], d["roofline"]["frac"], d["roofline"]["rows_scanned_per_step"], d["verify"])
print("second:", d.get("second_dataset"))
print("cpu:", d.get("cpu_baseline"))
PY
LYNSE_BENCH_ONLY_CONFIG=c4_share timeout 1200 python scripts/other_config.py c4_share > gpurun_out/t2/c4_share.json 2> gpurun_out/t2/c4_share.err; tail -2 gpurun_out/t2/c4_share.err; cat gpurun_out/t2/c4_share.json | head -c 3000
