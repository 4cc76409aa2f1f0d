#!/usr/bin/env bash
# Build libzkast.so for gfx950 (cross-compiles without a GPU).  Output: ../zkast/libzkast.so
# ZK_PROBES=1 additionally links ../zkast/libzkast_probes.so = the same objects + the tools/ probes (probe.hip and
# kernel variants kept for A/B timing); the product library never contains probe code.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../zkast"
LIBNAME="${ZK_LIB_NAME:-libzkast.so}"          # ZK_LIB_NAME / ZK_OBJ_DIR / ZK_EXTRA_FLAGS: experiment builds (tools/)
OBJ="${ZK_OBJ_DIR:-$HERE/build}"
mkdir -p "$OBJ"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# -amdgpu-mfma-vgpr-form: keep MFMA accumulators in arch VGPRs (gfx950 has a unified file); without it hipcc parks
# them in AGPRs and the attention softmax pays ~150 v_accvgpr_read/write per key tile.
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form"
PRODUCT="gemm gemm_c8 attention layernorm embed head logmel misc comm zkast"
PROBES="gemm_c8_v1 probe"
SRCS="$PRODUCT"
[ "${ZK_PROBES:-0}" = "1" ] && SRCS="$PRODUCT $PROBES"
pids=()
for f in $SRCS; do
  [ -f "$HERE/$f.hip" ] || continue
  if [ ! -f "$OBJ/$f.o" ] || [ "$HERE/$f.hip" -nt "$OBJ/$f.o" ] || [ "$HERE/zk_common.h" -nt "$OBJ/$f.o" ] || [ "$HERE/gemm_util.h" -nt "$OBJ/$f.o" ] \
     || [ "$HERE/../../include/zkast.h" -nt "$OBJ/$f.o" ]; then
    EXTRA=""
    # attention: scores are finite by construction (masking uses -1e30, not -inf), so fmax needs no sNaN-quieting
    # v_max x,x in front of every MFMA output (48 extra VALU per key tile otherwise)
    [ "$f" = "attention" ] && EXTRA="-fno-honor-nans"
    $HIPCC $FLAGS ${ZK_EXTRA_FLAGS:-} $EXTRA -c "$HERE/$f.hip" -o "$OBJ/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
objs=()
for f in $PRODUCT; do [ -f "$OBJ/$f.o" ] && objs+=("$OBJ/$f.o"); done
# -no-hip-rt: NO DT_NEEDED on libamdhip64 (and no RUNPATH into one ROCm tree).  The library binds to the HIP runtime the
# host process has loaded (zkast/lib.py::_ensure_hip_runtime loads exactly one, RTLD_GLOBAL; a C host links -lamdhip64
# itself).  With a hard-wired /opt/rocm runtime a process that also imports a PyTorch wheel (which
# ships its own, SONAME-less libamdhip64.so) ended up with two HIP + two HSA runtimes: "No HIP GPUs are available".
$HIPCC --offload-arch=gfx950 -shared -fPIC -no-hip-rt -o "$OUT/$LIBNAME" "${objs[@]}" -ldl ${ZK_LINK_LIBS:-}
echo "built $OUT/$LIBNAME"
if [ "${ZK_PROBES:-0}" = "1" ]; then
  pobjs=("${objs[@]}")
  for f in $PROBES; do [ -f "$OBJ/$f.o" ] && pobjs+=("$OBJ/$f.o"); done
  $HIPCC --offload-arch=gfx950 -shared -fPIC -no-hip-rt -o "$OUT/${ZK_PROBES_NAME:-libzkast_probes.so}" "${pobjs[@]}" -ldl ${ZK_LINK_LIBS:-}
  echo "built $OUT/${ZK_PROBES_NAME:-libzkast_probes.so}"
fi
