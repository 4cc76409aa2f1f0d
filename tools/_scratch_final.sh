cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4final
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r4final/tests.log 2>&1; echo tests rc $?; tail -2 gpurun_out/r4final/tests.log
( time timeout -k 10 400 python bench.py > gpurun_out/r4final/bench.json 2> gpurun_out/r4final/bench.err ) 2> gpurun_out/r4final/bench.time; echo bench rc $?; cat gpurun_out/r4final/bench.time
timeout -k 10 300 python bench.py --steps 20 --warmup 2 --no-cpu > gpurun_out/r4final/bench20.json 2> gpurun_out/r4final/bench20.err; echo bench20 rc $?
CLOCK_JSON=gpurun_out/r4final/gemm_clock.json timeout -k 10 300 python tools/gemm_stamps.py 512 st2 > gpurun_out/r4final/clock.txt 2>&1; cat gpurun_out/r4final/clock.txt
timeout -k 10 900 bash tools/profile_round.sh r04 > gpurun_out/r4final/profile.log 2>&1; echo prof rc $?
timeout -k 10 300 python tools/run_configs.py --config 3 > gpurun_out/r4final/config3.json 2> gpurun_out/r4final/config3.err; echo c3 rc $?
timeout -k 10 400 python tools/run_configs.py --config 4 > gpurun_out/r4final/config4.json 2> gpurun_out/r4final/config4.err; echo c4 rc $?
