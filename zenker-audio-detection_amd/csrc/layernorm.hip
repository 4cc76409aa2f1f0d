// LayerNorm(768, eps) over fp32 rows -> fp16 (hi[,lo]) planes for the next MFMA GEMM.
// Replaces nn.LayerNorm in ASTLayer ($TF/.../modeling_audio_spectrogram_transformer.py:199-200,216,222).
// One wave per row: 12 floats per lane (3 x dwordx4), two-pass statistics in registers, wave-shuffle reductions.
// HBM-bound: 3072 B read + 1536 (3072 with the lo plane) B written per row.
#include "zk_common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

#ifndef ZK_LN_WAVES
#define ZK_LN_WAVES 4      // rows (= waves) per workgroup
#endif
__global__ __launch_bounds__(64 * ZK_LN_WAVES) void layernorm_kernel(const float* __restrict__ x, int64_t row_stride,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int rows, half_t* o_hi,
                                                        half_t* o_lo, int lo_fmt, int32_t* rowexp, float eps, int rev, int tiled, int gather_tr) {
  const int lane = threadIdx.x & 63;
  int row = blockIdx.x * ZK_LN_WAVES + (threadIdx.x >> 6);
  if (row >= rows) return;
  if (rev) row = rows - 1 - row;
  int src_row = row;
  if (gather_tr) {      // compact row [b][f][t < gather_tr] <- token row b·1214 + 2 + f·101 + t
    const int npr = ZK_FOUT * gather_tr;
    const int b = row / npr, r = row - b * npr, f = r / gather_tr;
    src_row = b * ZK_SEQ + 2 + f * ZK_TOUT + (r - f * gather_tr);
  }
  const float* xr = x + (size_t)src_row * row_stride;
  f4_t v[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) v[i] = *(const f4_t*)(xr + i * 256 + lane * 4);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  const float mean = wave_sum(s) * (1.0f / ZK_HIDDEN);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d = v[i][j] - mean;
      q = fmaf(d, d, q);
    }
  const float var = wave_sum(q) * (1.0f / ZK_HIDDEN);
  const float rstd = 1.0f / sqrtf(var + eps);
  float y3[3][4];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = i * 256 + lane * 4;
    const f4_t g = *(const f4_t*)(gamma + c);
    const f4_t b = *(const f4_t*)(beta + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      y3[i][j] = fmaf((v[i][j] - mean) * rstd, g[j], b[j]);
      amax = fmaxf(amax, fabsf(y3[i][j]));
    }
  }
  float rs = 1.0f;      // row scale 2^-s of the ZK_F16C8 planes (zk_planes::rowexp)
  if (rowexp) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    const int sx = zk_row_exponent(amax);
    rs = ldexpf(1.0f, -sx);
    if (lane == 0) rowexp[row] = sx;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = i * 256 + lane * 4;
    h4_t hi;
    float y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      y[j] = y3[i][j] * rs;
      zk_pin(y[j]);
      hi[j] = (half_t)y[j];
    }
    // (tiled planes, zk_planes::tiled: the wave's 512 contiguous bytes become four whole 128-byte lines, one per 64-column chunk)
    const size_t oo = tiled ? zk_tiled_off(row, c, ZK_HIDDEN) : (size_t)row * ZK_HIDDEN + c;
    *(h4_t*)(o_hi + oo) = hi;
    if (o_lo) *(h4_t*)(o_lo + oo) = zk_lo4(y, hi, lo_fmt);
  }
}

}  // namespace

void zk_launch_layernorm(const float* x, int64_t row_stride, const float* gamma, const float* beta, int rows,
                         zk_planes out, float eps, hipStream_t s, int rev, int gather_tr) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(layernorm_kernel, dim3((rows + ZK_LN_WAVES - 1) / ZK_LN_WAVES), dim3(64 * ZK_LN_WAVES), 0, s, x, row_stride, gamma, beta, rows,
                     out.hi, out.lo, out.lo_fmt, (out.lo && out.lo_fmt == ZK_LO_C8) ? out.rowexp : nullptr, eps, rev, out.tiled, gather_tr);
}
