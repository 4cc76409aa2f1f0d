mkdir -p gpurun_out
timeout -k 10 200 python tools/gemm_stamps.py 512 st,stsym,stl1 > gpurun_out/r5p_stamps.log 2>&1
