# scratch: the command list of the last A/B run on the GPU box (gpurun -- 'bash tools/run_ab.sh'); edit freely
set -e
mkdir -p gpurun_out
timeout -k 10 420 python tools/gemm_ab_multi.py 512 5 r02,base,pf1,pf2,pf3,pf4 > gpurun_out/r3_pf_ab.log 2>&1
ZKAST_LIB=$PWD/zenker-audio-detection_amd/zkast/libzkast_pf2.so timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > gpurun_out/r3_pf2_ktests.log 2>&1
for v in r02 pf2 ""; do
  L=libzkast${v:+_$v}.so
  ZKAST_LIB=$PWD/zenker-audio-detection_amd/zkast/$L timeout -k 10 200 python bench.py --headline-only --steps 5 > gpurun_out/r3_bench_${v:-base}.json 2> gpurun_out/r3_bench_${v:-base}.err
done
