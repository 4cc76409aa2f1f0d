#!/usr/bin/env bash
# Build a probe variant of the library with extra -D flags for ONE kernel file (default gemm_c8):
#   tools/build_variant.sh <name> "<flags>" [file]   ->  zkast/libzkast_probes_<name>.so  (+ libzkast_<name>.so)
# Objects of the other files are copied from the main build directory, so only <file> is recompiled.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
CSRC="$ROOT/zenker-audio-detection_amd/csrc"
NAME="$1"; FLAGS="$2"; FILE="${3:-gemm_c8}"
OBJ="$CSRC/build_$NAME"
mkdir -p "$OBJ"
cp -p "$CSRC"/build/*.o "$OBJ"/ 2>/dev/null || true
rm -f "$OBJ/$FILE.o"
ZK_PROBES=1 ZK_OBJ_DIR="$OBJ" ZK_EXTRA_FLAGS="$FLAGS" ZK_LIB_NAME="libzkast_$NAME.so" ZK_PROBES_NAME="libzkast_probes_$NAME.so" \
  bash "$CSRC/build.sh"
