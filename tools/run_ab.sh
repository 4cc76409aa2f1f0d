set -e
Z=zenker-audio-detection_amd/zkast
timeout -k 10 300 python tools/attn_ab.py 512 > gpurun_out/nw.log 2>&1
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" > gpurun_out/nw_tests.log 2>&1
echo "== nw8" >> gpurun_out/nw.log
ZKAST_PROBES=$Z/libzkast_probes_nw8.so timeout -k 10 300 python tools/attn_ab.py 512 >> gpurun_out/nw.log 2>&1
ZKAST_LIB=$Z/libzkast_nw8.so timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" >> gpurun_out/nw_tests.log 2>&1
