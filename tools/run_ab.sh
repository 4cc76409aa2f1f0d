set -e
timeout -k 10 120 python tools/gemm_c6_ab.py 8 2 1 > gpurun_out/c6_small.log 2>&1
timeout -k 10 400 python tools/gemm_c6_ab.py 512 4 3 > gpurun_out/c6.log 2>&1
