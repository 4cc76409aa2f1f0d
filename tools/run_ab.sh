mkdir -p gpurun_out
timeout -k 10 300 python bench.py --headline-only --steps 5 > gpurun_out/r3j_bench.json 2> gpurun_out/r3j_bench.err
timeout -k 10 600 python bench.py --steps 3 > gpurun_out/r3j_bench_full.json 2> gpurun_out/r3j_bench_full.err
# N = 2 rehearsal on one GPU: RCCL refuses two ranks on one device -> the fallback must be reported, and REQUIRE_RCCL must fail
ZK_BENCH_ONE_GPU=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 1 --headline-only --batch 64 > gpurun_out/r3j_bench_n2.json 2> gpurun_out/r3j_bench_n2.err
ZK_BENCH_ONE_GPU=1 ZK_BENCH_REQUIRE_RCCL=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 1 --warmup 1 --headline-only --batch 64 > gpurun_out/r3j_bench_n2_req.json 2> gpurun_out/r3j_bench_n2_req.err
echo "require_rccl exit code $?" > gpurun_out/r3j_n2_req_rc.txt
