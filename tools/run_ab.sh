mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r3k_gpu_tests.log 2>&1
timeout -k 10 300 python bench.py --headline-only --steps 3 > gpurun_out/r3k_bench.out 2> gpurun_out/r3k_bench.err
ZK_BENCH_ONE_GPU=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 1 --warmup 1 --headline-only --batch 64 > gpurun_out/r3k_bench_n2.out 2> gpurun_out/r3k_bench_n2.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3k_smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r3k_smoke.log
