mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s -k "attention" > gpurun_out/r3f_attn_tests.log 2>&1
timeout -k 10 400 python tools/attn_ab_multi.py 512 5 qb1,base,qb2acc > gpurun_out/r3f_attn_ab.log 2>&1
