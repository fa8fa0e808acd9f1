# round 6: the exact-rescored ("tightened") thresholds of the selects against the coarse-rule thresholds (LYNSE_HIP_TIGHTEN=0), shard step and 10M step, alternating
S="timeout 300 python bench.py --no-cpu-baseline --no-configs --no-verify --steps 100 --warmup 5"
for round in 1 2; do for v in 1 0; do
  echo -n "TIGHTEN=$v shard(3 in flight) "; LYNSE_HIP_TIGHTEN=$v $S --rows 1250000 --in-flight 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['fallback_queries'], d['rescored_per_query'])"
  echo -n "TIGHTEN=$v 10M blocking       "; LYNSE_HIP_TIGHTEN=$v $S --steps 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['fallback_queries'], d['rescored_per_query'])"
done; done
