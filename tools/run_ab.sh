mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r5j_tests.log 2>&1 &&
timeout -k 10 400 python bench.py --steps 10 > gpurun_out/r5j_bench.json 2> gpurun_out/r5j_bench.err
