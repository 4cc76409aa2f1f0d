mkdir -p gpurun_out
timeout -k 10 400 python tools/gemm_ab_multi.py 512 5 sym,base,rp1,rp2,rt5 > gpurun_out/r5c_roles2.log 2>&1 &&
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r5c_tests.log 2>&1 &&
timeout -k 10 400 python bench.py --steps 10 > gpurun_out/r5c_bench.json 2> gpurun_out/r5c_bench.err
