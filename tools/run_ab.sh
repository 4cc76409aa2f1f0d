mkdir -p gpurun_out
OLD=$PWD/zenker-audio-detection_amd/zkast/libzkast_r3start.so
for i in 1 2 3; do
ZKAST_LIB=$OLD timeout -k 10 200 python bench.py --steps 8 --headline-only > gpurun_out/r3e_old_$i.json 2> gpurun_out/r3e_old_$i.err || exit 1
timeout -k 10 200 python bench.py --steps 8 --headline-only > gpurun_out/r3e_new_$i.json 2> gpurun_out/r3e_new_$i.err || exit 1
done
