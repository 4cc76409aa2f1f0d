mkdir -p gpurun_out
timeout -k 10 300 python tools/gemm_stamps.py 512 clk,clksym,clknoepi > gpurun_out/r5r_clock.log 2>&1
ZKP_ZERO=1 timeout -k 10 300 python tools/gemm_stamps.py 512 clk > gpurun_out/r5r_clock_zero.log 2>&1
