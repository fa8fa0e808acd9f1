for pe in 4 1000 10 4 1000 20; do
  timeout 300 python bench.py --no-cpu-baseline --no-configs --no-verify --steps 60 --warmup 5 --rows 1250000 --in-flight 3 --profile-every $pe 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('in-flight 3 profile-every', $pe, 'ms_per_step', d['ms_per_step'], 'blocking', d['blocking_ms_per_batch'], 'timed_steps', d['roofline'].get('timed_steps'))"
done
