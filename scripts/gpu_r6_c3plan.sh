for round in 1 2; do
for env in "X=1" "LYNSE_HIP_SAMPLE_GROWTH=64" "LYNSE_HIP_SAMPLE_GROWTH=8" "LYNSE_HIP_SAMPLE_ROWS_TO=32768" "LYNSE_HIP_SAMPLE_ROWS_TO=131072"; do
  echo -n "$env  "; env $env LAT_C3_ONLY=0 python scripts/r6_latency.py c3 2>/dev/null | grep config | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['raw_cabi_us'])"
done; done
