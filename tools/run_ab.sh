set -e
echo "== random data" > gpurun_out/zero.log
timeout -k 10 300 python tools/gemm_ab.py 512 2 4 3 >> gpurun_out/zero.log 2>&1
echo "== zero data" >> gpurun_out/zero.log
ZKP_ZERO=1 timeout -k 10 300 python tools/gemm_ab.py 512 2 4 3 >> gpurun_out/zero.log 2>&1
