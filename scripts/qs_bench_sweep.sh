#!/bin/bash
# bench.py ms/step for combinations of the scan variant and the stage growth of the sampled plan (same box, alternating)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/${1:-sweep}
for rep in 1 2; do
for cfg in "1 32" "3 32" "1 16" "1 8" "3 8" "0 32" "1 64"; do
  set -- $cfg
  r=$(LYNSE_HIP_QS=$1 LYNSE_HIP_SAMPLE_GROWTH=$2 python bench.py --gpus 1 --steps 30 --warmup 5 --no-configs --no-cpu-baseline --no-verify 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_us'], d['rescored_per_query'], d['roofline']['plan']['stages'])")
  echo "QS=$1 growth=$2 : ms/step, avg scan launch us, rescored/query, stages = $r"
done
done
