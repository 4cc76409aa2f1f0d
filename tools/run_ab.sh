set -e
Z=zenker-audio-detection_amd/zkast
ZKAST_PROBES=$Z/libzkast_probes_ntl.so AB_ONLY=o,fc2 timeout -k 10 300 python tools/gemm_ab.py 512 3 4 3 > gpurun_out/ntl.log 2>&1
AB_ONLY=o,fc2 timeout -k 10 300 python tools/gemm_ab.py 512 3 4 3 >> gpurun_out/ntl.log 2>&1
