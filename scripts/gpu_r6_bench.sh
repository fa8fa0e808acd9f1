# round 6: the default bench line + a digest of its new keys
mkdir -p gpurun_out/r6
T=${1:-bench_a}
python bench.py > gpurun_out/r6/$T.json 2> gpurun_out/r6/$T.err
tail -c 600 gpurun_out/r6/$T.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r6/$T.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], json.dumps(d["roofline"])[:1000])
print(json.dumps(d.get("cpu_baseline"))[:1600])
for k, v in d.get("configs", {}).items():
    print(k, json.dumps(v)[:1500])
PY
