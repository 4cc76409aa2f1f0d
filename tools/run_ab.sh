set -e
Z=zenker-audio-detection_amd/zkast
ZKAST_LIB=$Z/libzkast_lnr.so timeout -k 10 500 python bench.py --headline-only --steps 4 --warmup 2 > gpurun_out/lnr.log 2>&1
timeout -k 10 500 python bench.py --headline-only --steps 4 --warmup 2 > gpurun_out/lnr_base.log 2>&1
