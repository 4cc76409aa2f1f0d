set -e
Z=zenker-audio-detection_amd/zkast
timeout -k 10 300 python tools/attn_ab.py 512 > gpurun_out/attn.log 2>&1
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" > gpurun_out/attn_tests.log 2>&1
for v in top; do
  echo "== $v" >> gpurun_out/attn.log
  ZKAST_PROBES=$Z/libzkast_probes_$v.so timeout -k 10 300 python tools/attn_ab.py 512 >> gpurun_out/attn.log 2>&1
done
