mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s -k "attention" > gpurun_out/r3d_attn_tests.log 2>&1
ZKAST_LIB=$PWD/zenker-audio-detection_amd/zkast/libzkast_nw8novl.so timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s -k "attention" > gpurun_out/r3d_attn_tests_novl.log 2>&1
