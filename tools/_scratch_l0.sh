timeout -k 10 600 python -m pytest tests/test_model_gpu.py -x -q -k "layer0 or pruning or ragged or geometries or batch_properties" 2>&1 | tail -5
for round in 1 2; do
  for v in 0 1; do
    ZK_L0_REUSE=$v timeout -k 10 200 python bench.py --headline-only --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('reuse $v round $round', round(d['value'],1), 'win/s ', ' '.join(f\"{n} {k[n]['ms_per_launch']:.3f}x{k[n]['launches']}\" for n in ('gemm_qkv','gemm_patch','layernorm','embed')), 'exec GF/win/stage', round(d['roofline_end_to_end']['executed_gflop_per_window_stage'],2))"
  done
done
