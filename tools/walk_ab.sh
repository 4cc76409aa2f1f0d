#!/usr/bin/env bash
# FC1 tile-walk A/B (round 5): production walk (w0) against super-tile walks that keep the eight XCDs on WC
# column panels of W (w1 / w2 / w4 = -DZK_C8_WALK=WC), all four builds with -DZK_C8_STAMPS=2 (in-kernel clock), the production
# launch form (ZKP_TILED=1).  ms + GHz from tools/gemm_stamps.py (80 back-to-back launches per shape), FETCH_SIZE / WRITE_SIZE
# from separate rocprofv3 --pmc passes of the same command.  Run on the GPU box from the repo root.
set -uo pipefail
export TMPDIR=/tmp ZKP_TILED=1 ZKP_SHAPES=fc1,qkv
OUT=gpurun_out/walk_ab
mkdir -p "$OUT"
python3 tools/gemm_stamps.py 512 w0,w4,w2,w1,w0,w4,w2,w1 > "$OUT/stamps.txt" 2> "$OUT/stamps.err" || exit 1
for v in w0 w4 w2 w1; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --output-format csv -d "$OUT/pmc_${v}_$ctr" -- python3 tools/gemm_stamps.py 512 $v > /dev/null 2> "$OUT/pmc_${v}_$ctr.err" || exit 1
    f=$(find "$OUT/pmc_${v}_$ctr" -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$v" "$ctr" >> "$OUT/pmc.txt" <<'PY'
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
by = {}
for r in rows:
    if "gemm_c8_kernel" in r["Kernel_Name"]:
        k = "fc1(gelu)" if "gemm_c8_kernel<1" in r["Kernel_Name"] else ("qkv(store)" if "gemm_c8_kernel<0" in r["Kernel_Name"] else r["Kernel_Name"][:40])
        by.setdefault(k, []).append(float(r["Counter_Value"]))
for k, v in sorted(by.items()):
    print(f"{sys.argv[2]} {sys.argv[3]} {k}: median {statistics.median(v):.0f} KB per launch over {len(v)} launches")
PY
    rm -rf "$OUT/pmc_${v}_$ctr"
  done
done
cat "$OUT/stamps.txt" "$OUT/pmc.txt"
