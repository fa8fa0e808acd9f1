set -x
mkdir -p gpurun_out/r2
(N=1250000 timeout 250 python scripts/dbg_fs.py; N=2400000 timeout 250 python scripts/dbg_fs.py) 2>&1 | grep -E "==|lost rows|top rows|^fused|^unfused" > gpurun_out/r2/dbg.log
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2/tests.log
B="timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 5"
$B > gpurun_out/r2/b10m_fused.json 2> gpurun_out/r2/b10m_fused.err
LYNSE_HIP_FUSED_SAMPLE=0 LYNSE_HIP_FUSED_TAIL=0 $B > gpurun_out/r2/b10m_unfused.json 2> gpurun_out/r2/b10m_unfused.err
S="$B --rows 1250000 --in-flight 3 --no-verify"
LYNSE_BENCH_FORCE_COMM=1 $S > gpurun_out/r2/s_fused.json 2> gpurun_out/r2/s_fused.err
LYNSE_BENCH_FORCE_COMM=1 LYNSE_HIP_FUSED_SAMPLE=0 LYNSE_HIP_FUSED_TAIL=0 $S > gpurun_out/r2/s_unfused.json 2> gpurun_out/r2/s_unfused.err
LYNSE_BENCH_FORCE_COMM=1 LYNSE_HIP_FUSED_SAMPLE=1 LYNSE_HIP_FUSED_TAIL=0 $S > gpurun_out/r2/s_fs_only.json 2> gpurun_out/r2/s_fs_only.err
LYNSE_BENCH_FORCE_COMM=1 LYNSE_HIP_FUSED_SAMPLE=0 LYNSE_HIP_FUSED_TAIL=1 $S > gpurun_out/r2/s_ft_only.json 2> gpurun_out/r2/s_ft_only.err
$B --rows 1250000 --in-flight 1 > gpurun_out/r2/s_blocking_fused.json 2> gpurun_out/r2/s_blocking_fused.err
LYNSE_HIP_FUSED_SAMPLE=0 LYNSE_HIP_FUSED_TAIL=0 $B --rows 1250000 --in-flight 1 > gpurun_out/r2/s_blocking_unfused.json 2> gpurun_out/r2/s_blocking_unfused.err
cat gpurun_out/r2/dbg.log
cat gpurun_out/r2/tests.log
for f in gpurun_out/r2/*.json; do echo $f; python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1])
    print(d['ms_per_step'], d['value'], 'blocking', d['blocking_ms_per_batch'], 'pipe', d['pipeline_us_per_step'], 'resc', d['rescored_per_query'], 'fb', d['fallback_queries'], 'l/s', d['roofline']['launches_per_step'], 'avg_us', d['roofline']['avg_launch_us'], 'i8', d['roofline']['plan']['int8_coarse_pass'], 'fs', d['roofline']['plan']['fused_sample_stage'], (d.get('verify') or {}).get('recall_at_k'))
except Exception as e: print('ERR', e)
"; done
