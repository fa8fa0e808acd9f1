# round 6: the sample size of the 1.25M-row shard step (65,536 rows = 5 % of the shard by default): LYNSE_HIP_SAMPLE_ROWS_TO, 3 in flight, alternating
S="timeout 300 python bench.py --no-cpu-baseline --no-configs --no-verify --steps 120 --warmup 5 --rows 1250000 --in-flight 3"
for round in 1 2; do for v in 65536 49152 32768 16384; do
  echo -n "SAMPLE_ROWS_TO=$v  "; LYNSE_HIP_SAMPLE_ROWS_TO=$v $S 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['fallback_queries'], d['rescored_per_query'], d['roofline']['launches_per_step'])"
done; done
