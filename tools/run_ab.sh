mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r3f_tests.log 2>&1 || exit 1
bash tools/profile_round.sh r03f > gpurun_out/r3f_profile.log 2>&1 &&
timeout -k 10 400 python bench.py --steps 20 > gpurun_out/r3f_bench20.json 2> gpurun_out/r3f_bench20.err &&
timeout -k 10 300 python tools/run_configs.py --config 3 > gpurun_out/r3f_cfg3.json 2> gpurun_out/r3f_cfg3.err &&
timeout -k 10 400 python tools/run_configs.py --config 4 > gpurun_out/r3f_cfg4.json 2> gpurun_out/r3f_cfg4.err
