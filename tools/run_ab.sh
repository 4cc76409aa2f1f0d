set -e
AB_ONLY=o,fc2 timeout -k 10 300 python tools/gemm_ab.py 512 3 4 3 >> gpurun_out/sr.log 2>&1
AB_ONLY=o,fc2 timeout -k 10 300 python tools/gemm_ab.py 256 3 4 3 >> gpurun_out/sr.log 2>&1
AB_ONLY=o,fc2 timeout -k 10 300 python tools/gemm_ab.py 107 3 4 3 >> gpurun_out/sr.log 2>&1
AB_ONLY=o,fc2 timeout -k 10 300 python tools/gemm_ab.py 17 3 4 3 >> gpurun_out/sr.log 2>&1
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > gpurun_out/sr_tests.log 2>&1
