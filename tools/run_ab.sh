mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 5 --headline-only --micro-batch 1024 > gpurun_out/r5l_mb1024.json 2> gpurun_out/r5l_mb1024.err &&
timeout -k 10 300 python bench.py --steps 5 --headline-only --micro-batch 512 > gpurun_out/r5l_mb512.json 2> gpurun_out/r5l_mb512.err
