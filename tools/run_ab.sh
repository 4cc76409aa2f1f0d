mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -m gpu -q > gpurun_out/r5w_tests.log 2>&1
