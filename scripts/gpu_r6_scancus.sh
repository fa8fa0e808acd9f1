# round 6: the 1.25M-row shard step with 3 batches in flight against the number of CUs the persistent threshold-stage scan takes (LYNSE_HIP_SCAN_CUS):
# do a few free CUs let the other batches' short kernels run beside the scan?  Alternating, same box.
S="timeout 300 python bench.py --no-cpu-baseline --no-configs --no-verify --steps 120 --warmup 5 --rows 1250000 --in-flight 3"
for round in 1 2; do
for c in 256 252 248 240 224; do
  echo -n "SCAN_CUS=$c  "; LYNSE_HIP_SCAN_CUS=$c $S 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done
