#!/usr/bin/env bash
# Build a probe variant of the library with extra -D flags for ONE kernel file (default gemm_c8):
#   tools/build_variant.sh <name> "<flags>" [file] [alternative source]  ->  zkast/libzkast_probes_<name>.so
# Objects of the other files are copied from the main build directory, so only <file> is recompiled; an alternative
# source (e.g. tools/archive/*.hip.txt) is compiled in place of csrc/<file>.hip.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
CSRC="$ROOT/zenker-audio-detection_amd/csrc"
NAME="$1"; FLAGS="$2"; FILE="${3:-gemm_c8}"; ALT="${4:-}"
OBJ="$CSRC/build_$NAME"
mkdir -p "$OBJ"
cp -p "$CSRC"/build/*.o "$OBJ"/ 2>/dev/null || true
if [ -n "$ALT" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form \
    -I"$CSRC" $FLAGS -x hip -c "$ALT" -o "$OBJ/$FILE.o"
  touch "$OBJ/$FILE.o"
else
  rm -f "$OBJ/$FILE.o"
fi
ZK_PROBES=1 ZK_OBJ_DIR="$OBJ" ZK_EXTRA_FLAGS="$FLAGS" ZK_LIB_NAME="libzkast_$NAME.so" ZK_PROBES_NAME="libzkast_probes_$NAME.so" \
  bash "$CSRC/build.sh"
