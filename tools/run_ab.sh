mkdir -p gpurun_out
timeout -k 10 900 bash tools/profile_round.sh r03 > gpurun_out/r3x_profile.log 2>&1
timeout -k 10 900 python bench.py --steps 20 --warmup 1 > gpurun_out/r3x_bench20.json 2> gpurun_out/r3x_bench20.err
timeout -k 10 300 python tools/run_configs.py --config 3 > gpurun_out/r3x_cfg3.json 2> gpurun_out/r3x_cfg3.err
timeout -k 10 300 python tools/run_configs.py --config 4 > gpurun_out/r3x_cfg4.json 2> gpurun_out/r3x_cfg4.err
