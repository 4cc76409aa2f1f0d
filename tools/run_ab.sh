set -e
Z=zenker-audio-detection_amd/zkast
for v in ss2 ss5; do
  echo "== $v" >> gpurun_out/ss.log
  ZKAST_PROBES=$Z/libzkast_probes_$v.so AB_ONLY=qkv,fc1 timeout -k 10 300 python tools/gemm_ab.py 512 3 4 3 >> gpurun_out/ss.log 2>&1
done
