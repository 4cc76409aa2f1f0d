#!/usr/bin/env bash
# Round-end measurement set on ONE box (run from the repo root on the GPU box): rocprofv3 kernel stats + PMC passes of the headline
# region, the in-kernel clocks of a stamp build, the default bench line, the two-rank rehearsal of the N > 1 code path.
set -uo pipefail
TAG="${1:-r05}"
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/profile_round.sh "$TAG" > "gpurun_out/${TAG}_profile_round.log" 2>&1 || { tail -5 "gpurun_out/${TAG}_profile_round.log"; exit 1; }
echo "profile round done"
# (needs the stamp build: bash tools/build_variant.sh s2 "-DZK_C8_STAMPS=2")
ZKP_TILED=1 CLOCK_JSON="gpurun_out/${TAG}_gemm_clock.json" python3 tools/gemm_stamps.py 512 s2 > "gpurun_out/${TAG}_gemm_clock.txt" 2>&1 || { tail -5 "gpurun_out/${TAG}_gemm_clock.txt"; exit 1; }
echo "clock done"
cp "gpurun_out/prof_${TAG}/pmc_traffic.json" "profiles/${TAG}_pmc_traffic.json"; cp "gpurun_out/${TAG}_gemm_clock.json" "profiles/${TAG}_gemm_clock.json"
python3 bench.py > "gpurun_out/${TAG}_bench.json" 2> "gpurun_out/${TAG}_bench.err" || { tail -5 "gpurun_out/${TAG}_bench.err"; exit 1; }
echo "bench done"
ZK_BENCH_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --steps 2 --warmup 1 --headline-only > "gpurun_out/${TAG}_bench_n2_rehearsal.json" 2> "gpurun_out/${TAG}_bench_n2.err" || tail -5 "gpurun_out/${TAG}_bench_n2.err"
echo "n2 rehearsal done"
python3 tools/run_configs.py --config 3 > "gpurun_out/${TAG}_config3_n1.json" 2> "gpurun_out/${TAG}_config3.err" || tail -3 "gpurun_out/${TAG}_config3.err"
python3 -c "
import json
d = json.load(open('gpurun_out/${TAG}_bench.json'))
print('headline', round(d['value'], 1), d['dtype'], 'roofline', {k: d['roofline'].get(k) for k in ('kernel', 'frac', 'traffic', 'mfma_util', 'in_kernel_clock_ghz')})
print('legs', d['legs'])
print('parity', d.get('parity_in_bench'))
print('cpu', d.get('cpu_baseline', {}).get('value'))
"
