# scratch: the command list of the last A/B run on the GPU box (gpurun -- 'bash tools/run_ab.sh'); edit freely
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1
