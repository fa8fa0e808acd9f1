#!/bin/bash
cd "$(dirname "$0")/.."
for rep in 1 2; do
for cfg in "256 1" "256 2" "248 2" "240 2" "240 3" "232 3" "224 3" "240 4"; do
  set -- $cfg
  r=$(LYNSE_HIP_SCAN_CUS=$1 LYNSE_HIP_CONTEXTS=4 python bench.py --gpus 1 --steps 40 --warmup 8 --in-flight $2 --no-configs --no-cpu-baseline --no-verify 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['blocking_ms_per_batch'], d['config']['batches_in_flight'])")
  echo "scan CUs=$1 in-flight=$2 : ms/step, blocking ms/batch, in flight = $r"
done
done
