set -e
Z=zenker-audio-detection_amd/zkast
echo "== production (gshift 0)" >> gpurun_out/abgs.log
timeout -k 10 300 python tools/gemm_ab.py 512 3 4 3 >> gpurun_out/abgs.log 2>&1
for v in gs1 gs2 gs3 gs4; do
echo "== $v" >> gpurun_out/abgs.log
ZKAST_PROBES=$Z/libzkast_probes_$v.so timeout -k 10 300 python tools/gemm_ab.py 512 3 4 3 >> gpurun_out/abgs.log 2>&1
done
