set -e
AB_ONLY=fc1 timeout -k 10 300 python tools/gemm_ab.py 512 3 4 5 > gpurun_out/st.log 2>&1
AB_ONLY=fc1 timeout -k 10 300 python tools/gemm_ab.py 512 3 4 5 >> gpurun_out/st.log 2>&1
