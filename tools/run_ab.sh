mkdir -p gpurun_out
bash tools/profile_round.sh r03e > gpurun_out/r3e_profile.log 2>&1 &&
timeout -k 10 400 python bench.py --steps 20 > gpurun_out/r3e_bench20.json 2> gpurun_out/r3e_bench20.err &&
timeout -k 10 300 python tools/run_configs.py --config 3 > gpurun_out/r3e_cfg3.json 2> gpurun_out/r3e_cfg3.err &&
timeout -k 10 400 python tools/run_configs.py --config 4 > gpurun_out/r3e_cfg4.json 2> gpurun_out/r3e_cfg4.err
