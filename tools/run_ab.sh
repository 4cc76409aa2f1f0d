set -e
Z=zenker-audio-detection_amd/zkast
ZKAST_PROBES=$Z/libzkast_probes_xnt.so timeout -k 10 300 python tools/gemm_ab.py 512 3 4 3 > gpurun_out/xnt.log 2>&1
timeout -k 10 300 python tools/gemm_ab.py 512 3 4 3 >> gpurun_out/xnt.log 2>&1
