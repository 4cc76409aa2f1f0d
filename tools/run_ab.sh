set -e
timeout -k 10 400 python tools/gemm_ab.py 512 3 4 5 > gpurun_out/ab512.log 2>&1
timeout -k 10 200 python tools/gemm_ab.py 65 3 2 2 > gpurun_out/ab65.log 2>&1
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > gpurun_out/t_gemm.log 2>&1
