for pe in 4 10 20 1; do
  timeout 300 python bench.py --no-cpu-baseline --no-configs --no-verify --steps 40 --warmup 5 --profile-every $pe 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('profile-every', $pe, 'ms_per_step', d['ms_per_step'], 'blocking', d['blocking_ms_per_batch'], 'frac', d['roofline']['frac'], 'launches', d['roofline'].get('launches_timed'))"
done
