#!/usr/bin/env bash
# Build libzkast.so for gfx950 (cross-compiles without a GPU).  Output: ../zkast/libzkast.so
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../zkast"
OBJ="${ZK_OBJ_DIR:-$HERE/build}"
mkdir -p "$OBJ"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
pids=()
for f in gemm attention layernorm embed head logmel misc zkast; do
  if [ ! -f "$OBJ/$f.o" ] || [ "$HERE/$f.hip" -nt "$OBJ/$f.o" ] || [ "$HERE/zk_common.h" -nt "$OBJ/$f.o" ] \
     || [ "$HERE/../../include/zkast.h" -nt "$OBJ/$f.o" ]; then
    $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$OBJ/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libzkast.so" "$OBJ"/{gemm,attention,layernorm,embed,head,logmel,misc,zkast}.o
echo "built $OUT/libzkast.so"
