for cfg in "20 60" "60 60" "60 0" "60 60" "120 60" "60 200"; do
  set -- $cfg
  timeout 300 python bench.py --no-cpu-baseline --no-configs --no-verify --steps $1 --warmup 5 --rows 1250000 --in-flight 3 --settle-ms $2 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('steps $1 settle $2', 'ms_per_step', d['ms_per_step'], 'blocking', d['blocking_ms_per_batch'], 'pipeline_us', d['pipeline_us_per_step'], 'settle_steps', d['config']['settle_steps'])"
done
