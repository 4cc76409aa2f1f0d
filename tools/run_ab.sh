set -e
AB_ONLY=fc1 timeout -k 10 300 python tools/gemm_ab.py 512 3 4 3 > gpurun_out/gelu.log 2>&1
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > gpurun_out/gelu_tests.log 2>&1
