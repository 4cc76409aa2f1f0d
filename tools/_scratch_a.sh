timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_batch_gpu.py -x -q 2>&1 | tail -3
for i in 1 2; do timeout -k 10 200 python bench.py --headline-only --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print(round(d['value'],1), 'win/s ', ' '.join(f\"{n} {k[n]['ms_per_launch']:.3f}x{k[n]['launches']}\" for n in ('attention','gemm_qkv','embed','gemm_fc1')))"; done
