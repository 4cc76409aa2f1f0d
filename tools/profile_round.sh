#!/usr/bin/env bash
# rocprofv3 passes of the bench's headline region on the GPU box (run from the repo root):
#   kernel trace + stats, then one --pmc pass each for FETCH_SIZE, WRITE_SIZE and the SQ busy counters (never combined
#   with a trace domain).  Summaries land in gpurun_out/prof_<tag>/; copy what is to be judged into profiles/.
set -euo pipefail
TAG="${1:-r02}"
export TMPDIR=/tmp
OUT="gpurun_out/prof_$TAG"
mkdir -p "$OUT"
B="bench.py --steps 2 --warmup 1 --headline-only"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 $B > "$OUT/bench_kt.json" 2> "$OUT/kt.err"
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 $B > /dev/null 2> "$OUT/fetch.err"
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 $B > /dev/null 2> "$OUT/write.err"
echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -- python3 $B > /dev/null 2> "$OUT/sq.err"
echo "sq done"
STATS=$(find "$OUT/kt" -name "*kernel_stats.csv" | head -1)
cp "$STATS" "$OUT/kernel_stats.csv"
F=$(find "$OUT/fetch" -name "*counter_collection.csv" | head -1)
W=$(find "$OUT/write" -name "*counter_collection.csv" | head -1)
S=$(find "$OUT/sq" -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py "$OUT/pmc_traffic.json" "bench.py --steps 2 --warmup 1 --headline-only (1024 windows, g=1.0, f16c8), rocprofv3 --pmc, separate passes" "$F" "$W" "$S" > "$OUT/pmc_summary.txt"
# keep the merged scratch small: drop the raw per-dispatch csv files
rm -rf "$OUT/kt" "$OUT/fetch" "$OUT/write" "$OUT/sq"
head -25 "$OUT/kernel_stats.csv"
