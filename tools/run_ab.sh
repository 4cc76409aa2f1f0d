# scratch: the command list of the last A/B run on the GPU box (gpurun -- 'bash tools/run_ab.sh'); edit freely
set -e
timeout -k 10 300 python tools/gemm_ab.py 512 3 4 3 > gpurun_out/ab.log 2>&1
