mkdir -p gpurun_out
timeout -k 10 400 python tools/gemm_ab_multi.py 512 5 base,pf3d2,pf3d4,sym,pf3d3sym > gpurun_out/r5e_pf3.log 2>&1
