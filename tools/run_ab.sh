mkdir -p gpurun_out
for mb in 0 341 256 171 128 107 64 35 17 8; do
timeout -k 10 200 python bench.py --headline-only --steps 3 --micro-batch $mb > gpurun_out/r3i_mb_$mb.json 2> gpurun_out/r3i_mb_$mb.err
done
