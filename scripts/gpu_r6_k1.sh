# round 6: the fused IVF few-query launch and the warmed fused FLAT search: latency A/B + the tests that cover them
python scripts/r6_latency.py c1 c4 2>&1 | grep config
LYNSE_HIP_IVF_FUSE=0 python scripts/r6_latency.py c4 2>&1 | grep config
python -m pytest tests/test_gpu_ivf_parity.py tests/test_gpu_concurrent_readers.py tests/test_gpu_collection_glue.py tests/test_gpu_storage_formats.py -x -q 2>&1 | tail -4
python -m pytest tests/test_gpu_flat_parity.py -x -q -k "fused or nan" 2>&1 | tail -3
python -m pytest tests/test_gpu_baseline_configs.py -x -q -k "c4" 2>&1 | tail -3
