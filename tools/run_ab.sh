mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -s 2>&1 | grep -h "max-abs logit err\|passed\|failed\|FAILED" > gpurun_out/r3z_tests.log
timeout -k 10 500 python tools/gemm_ab_multi.py 512 5 old,base > gpurun_out/r3z_gemm_ab.log 2>&1
timeout -k 10 300 python bench.py --headline-only --steps 5 > gpurun_out/r3z_bench_new.json 2> gpurun_out/r3z_bench_new.err
ZKAST_LIB=$PWD/zenker-audio-detection_amd/zkast/libzkast_old.so timeout -k 10 300 python bench.py --headline-only --steps 5 > gpurun_out/r3z_bench_old.json 2> gpurun_out/r3z_bench_old.err
