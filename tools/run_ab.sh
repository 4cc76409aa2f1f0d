mkdir -p gpurun_out
ZKAST_LIB=$PWD/zenker-audio-detection_amd/zkast/libzkast_per2.so timeout -k 10 200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention" > gpurun_out/r5n_tests.log 2>&1 &&
timeout -k 10 300 python tools/attn_ab_multi.py 512 7 prev,per0,per2 > gpurun_out/r5n_att_ab.log 2>&1
