mkdir -p gpurun_out
AB_CHECK=0 timeout -k 10 400 python tools/gemm_ab_multi.py 512 5 base,prio_g > gpurun_out/r3l_prio_gemm.log 2>&1
timeout -k 10 300 python tools/attn_ab_multi.py 512 5 base,prio_a > gpurun_out/r3l_prio_attn.log 2>&1
