mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_batch_gpu.py tests/test_model_gpu.py -m gpu -q -k "cache or batch or cascade" > gpurun_out/r3p_tests.log 2>&1
