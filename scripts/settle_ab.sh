for sm in 60 0 60 0; do
  timeout 300 python bench.py --no-cpu-baseline --no-configs --no-verify --steps 20 --warmup 5 --settle-ms $sm 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('settle', $sm, 'N=1 ms_per_step', d['ms_per_step'], 'blocking', d['blocking_ms_per_batch'], 'frac', d['roofline']['frac'], d['config']['settle_steps'])"
done
for sm in 60 0 60 0; do
  timeout 300 python bench.py --no-cpu-baseline --no-configs --no-verify --steps 20 --warmup 5 --rows 1250000 --in-flight 3 --settle-ms $sm 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('settle', $sm, 'shard ms_per_step', d['ms_per_step'], 'blocking', d['blocking_ms_per_batch'], d['config']['settle_steps'])"
done
