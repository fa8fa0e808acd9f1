#!/bin/bash
# round 5: do the search contexts' streams share one hardware queue?  shard step (3 in flight) against GPU_MAX_HW_QUEUES, and the queue / stream ids of a trace
mkdir -p gpurun_out/queues
B="python bench.py --no-cpu-baseline --no-verify --no-configs --rows 1250000 --steps 80 --warmup 5 --in-flight 3"
for r in 1 2; do for q in 1 2 4 8; do
  GPU_MAX_HW_QUEUES=$q timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('GPU_MAX_HW_QUEUES=$q ms_per_step', d['ms_per_step'])"
done; done
export TMPDIR=/tmp; ROOT=$(pwd)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/queues/t -o u --output-format csv -- bash -c "cd $ROOT && $B --steps 20" > $ROOT/gpurun_out/queues/t.log 2>&1)
f=$(find gpurun_out/queues/t -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'lynse::' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
c=collections.Counter((r['Queue_Id'],r['Stream_Id']) for r in rows)
print('(queue, stream) -> dispatches', dict(c))
w=rows[-24:]
t0=int(w[0]['Start_Timestamp']); pe=None
for r in w:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print(r['Kernel_Name'].replace('void ','').replace('lynse::','')[:30].ljust(30),'q',r['Queue_Id'],'s',r['Stream_Id'],round((s-t0)/1e3,1),round((e-s)/1e3,1),'' if pe is None else round((s-pe)/1e3,1))
    pe=e if pe is None else max(pe,e)
PY
find gpurun_out -name "*kernel_trace.csv" -size +3M -delete
