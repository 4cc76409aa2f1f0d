mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -m gpu -q -k "features_set or compact_cache or feature_cache" > gpurun_out/r3o_tests.log 2>&1
