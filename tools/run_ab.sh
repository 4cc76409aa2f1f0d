mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s -k "attention" > gpurun_out/r3n_attn_tests.log 2>&1
