// Patch-embedding front end: im2col of the (time=1024, mel=128) spectrogram into the A operand of the
// [B*1212, 256] x [256, 768] patch GEMM, and the cls / distillation rows.
// Replaces ASTPatchEmbeddings / ASTEmbeddings ($TF/.../modeling_audio_spectrogram_transformer.py:57-61,89-99):
//   Conv2d(1, 768, 16x16, stride (10,10)) over x[b, 0, freq, time]; token = f*101 + t; K index = kf*16 + kt.
//
// Two sources:
//  * compact: un-normalised log-mel rows [N, n_frames, 128] written by logmel.hip; rows >= n_frames are the
//    extractor's 0.0 padding; (x - mean) / (2 std) is applied here in fp32 exactly as ASTFeatureExtractor.normalize
//    ($TF/.../feature_extraction_audio_spectrogram_transformer.py:157-158,228-229).  A window index list lets
//    stage 2 gather the gated windows without copying features.
//  * full: caller-provided, already normalised input_values (B, 1024, 128) — the HF model contract.
#include "zk_common.h"

namespace {

// one thread = 8 consecutive K elements (fixed kf, kt0 = 0 or 8) of one patch row
template <bool COMPACT>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ src, int n_frames,
                                                     const int32_t* __restrict__ win_idx, int n_windows, float mean,
                                                     float std2, half_t* __restrict__ o_hi, half_t* __restrict__ o_lo, int lo_fmt,
                                                     int t_real) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int npr = t_real ? ZK_FOUT * t_real : ZK_NPATCH, tpf = t_real ? t_real : ZK_TOUT;      // rows per window, patches per f
  const int64_t total = (int64_t)n_windows * npr * 32;
  if (gid >= total) return;
  const int seg = (int)(gid & 31);            // 32 segments of 8 per 256-wide row
  const int64_t prow = gid >> 5;              // b*npr + f*tpf + t
  const int b = (int)(prow / npr);
  const int p = (int)(prow - (int64_t)b * npr);
  const int f = p / tpf, t = p - f * tpf;
  const int kf = seg >> 1, kt0 = (seg & 1) * 8;
  const int mel = f * ZK_FSTRIDE + kf;
  const int time0 = t * ZK_TSTRIDE + kt0;
  const int w = COMPACT ? (win_idx ? win_idx[b] : b) : b;
  h8_t hi;
  float vv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int time = time0 + j;
    float v;
    if constexpr (COMPACT) {
      const float raw = time < n_frames ? src[((size_t)w * n_frames + time) * ZK_NMEL + mel] : 0.0f;
      v = (raw - mean) / std2;
    } else {
      v = src[((size_t)w * ZK_MAXLEN + time) * ZK_NMEL + mel];
    }
    zk_pin(v);
    hi[j] = (half_t)v;
    vv[j] = v;
  }
  *(h8_t*)(o_hi + prow * ZK_PATCH_K + seg * 8) = hi;
  if (o_lo) {
    *(h4_t*)(o_lo + prow * ZK_PATCH_K + seg * 8) = zk_lo4(vv, __builtin_shufflevector(hi, hi, 0, 1, 2, 3), lo_fmt);
    *(h4_t*)(o_lo + prow * ZK_PATCH_K + seg * 8 + 4) = zk_lo4(vv + 4, __builtin_shufflevector(hi, hi, 4, 5, 6, 7), lo_fmt);
  }
}

__global__ __launch_bounds__(256) void cls_rows_kernel(float* __restrict__ hidden, const float* __restrict__ cls,
                                                       const float* __restrict__ dist,
                                                       const float* __restrict__ pos, int n_windows) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= n_windows * 2 * ZK_HIDDEN) return;
  const int c = gid % ZK_HIDDEN;
  const int r = (gid / ZK_HIDDEN) & 1;
  const int b = gid / (2 * ZK_HIDDEN);
  hidden[((size_t)b * ZK_SEQ + r) * ZK_HIDDEN + c] = (r == 0 ? cls[c] : dist[c]) + pos[r * ZK_HIDDEN + c];
}

// rows 0 (cls) and 1 (distillation) of every window -> compact [2*n_windows, 768] (attention output planes and the
// fp32 residual stream): the only rows the head consumes, so the last layer's O / MLP GEMMs run on these alone.
__global__ __launch_bounds__(256) void gather_tok01_kernel(const half_t* __restrict__ a_hi, const half_t* __restrict__ a_lo,
                                                           const float* __restrict__ hidden, int n_windows,
                                                           half_t* __restrict__ o_hi, half_t* __restrict__ o_lo,
                                                           float* __restrict__ h_out, int a_tiled) {
  const int gid = blockIdx.x * 256 + threadIdx.x;       // one thread = 4 channels of one row
  if (gid >= n_windows * 2 * (ZK_HIDDEN / 4)) return;
  const int c4 = gid % (ZK_HIDDEN / 4);
  const int r = gid / (ZK_HIDDEN / 4);                   // b*2 + tok
  const size_t src = ((size_t)(r >> 1) * ZK_SEQ + (r & 1)) * ZK_HIDDEN + c4 * 4;
  const size_t dst = (size_t)r * ZK_HIDDEN + c4 * 4;
  // (the attention planes may lie in k-slice-major tiles, zk_planes::tiled; the compact output is row-major)
  const size_t asrc = a_tiled ? zk_tiled_off((r >> 1) * ZK_SEQ + (r & 1), c4 * 4, ZK_HIDDEN) : src;
  *(h4_t*)(o_hi + dst) = *(const h4_t*)(a_hi + asrc);
  if (o_lo) *(h4_t*)(o_lo + dst) = *(const h4_t*)(a_lo + asrc);
  *(f4_t*)(h_out + dst) = *(const f4_t*)(hidden + src);
}

// ---- last-layer pruning, query side (zkast.hip: forward_micro) --------------------------------------------------------------
// The pruned last layer needs q for tokens 0/1 only.  It is computed for the first ZK_QROWS = 32 tokens of every window — the
// rows ONE attention wave owns: a wave's deferred-rescale decision is wave-uniform, so tokens 0/1 keep their exact bits only
// if their 30 wave-mates carry the same q as in the unpruned forward.  Their LayerNorm rows are gathered into compact planes
// (with their row exponents), a 32-rows-per-window GEMM computes q, and the rows go back to columns 0..767 of the window's
// q|k|v planes.  one thread = 8 channels of one row
__global__ __launch_bounds__(256) void gather_xq_kernel(const half_t* __restrict__ x_hi, const half_t* __restrict__ x_lo,
                                                        const int32_t* __restrict__ x_exp, int x_tiled, int n_windows,
                                                        half_t* __restrict__ o_hi, half_t* __restrict__ o_lo,
                                                        int32_t* __restrict__ o_exp) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= n_windows * ZK_QROWS * (ZK_HIDDEN / 8)) return;
  const int c8 = gid % (ZK_HIDDEN / 8);
  const int r = gid / (ZK_HIDDEN / 8);                   // b*ZK_QROWS + tok
  const int srow = (r / ZK_QROWS) * ZK_SEQ + (r % ZK_QROWS);
  const size_t src = x_tiled ? zk_tiled_off(srow, c8 * 8, ZK_HIDDEN) : (size_t)srow * ZK_HIDDEN + c8 * 8;
  const size_t dst = (size_t)r * ZK_HIDDEN + c8 * 8;
  *(h8_t*)(o_hi + dst) = *(const h8_t*)(x_hi + src);
  if (o_lo) *(h8_t*)(o_lo + dst) = *(const h8_t*)(x_lo + src);
  if (o_exp && c8 == 0) o_exp[r] = x_exp ? x_exp[srow] : 0;
}

__global__ __launch_bounds__(256) void scatter_q_kernel(const half_t* __restrict__ q_hi, const half_t* __restrict__ q_lo,
                                                        int n_windows, half_t* __restrict__ o_hi, half_t* __restrict__ o_lo) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= n_windows * ZK_QROWS * (ZK_HIDDEN / 8)) return;
  const int c8 = gid % (ZK_HIDDEN / 8);
  const int r = gid / (ZK_HIDDEN / 8);
  const size_t src = (size_t)r * ZK_HIDDEN + c8 * 8;
  const size_t dst = ((size_t)(r / ZK_QROWS) * ZK_SEQ + (r % ZK_QROWS)) * (3 * ZK_HIDDEN) + c8 * 8;
  *(h8_t*)(o_hi + dst) = *(const h8_t*)(q_hi + src);
  if (o_lo) *(h8_t*)(o_lo + dst) = *(const h8_t*)(q_lo + src);
}

// ---- layer-0 constant-row reuse (zkast.hip: build_l0_table) ---------------------------------------------------------------
// Of a window's 1214 token rows only the 12·t_real patch tokens with t < t_real see real frames (t_real = ceil(n_frames / 10)
// = 10 for a 1 s window); cls, distillation and the other patch tokens have a layer-0 input that is the same for every
// window.  Their residual rows and their layer-0 q|k|v rows come from a table, bit-identical to computing them.
// one workgroup = one constant row of ZK_L0_WPB consecutive windows (192 threads x 4 channels): read once, stored 8 times
__global__ __launch_bounds__(192) void l0_fill_hidden_kernel(float* __restrict__ hidden, const float* __restrict__ table,
                                                             int n_windows, int t_real) {
  const int tpad = ZK_TOUT - t_real, nconst = 2 + ZK_FOUT * tpad;
  const int i = blockIdx.x % nconst, b0 = (blockIdx.x / nconst) * 8;
  int row = i;
  if (i >= 2) { const int j = i - 2, f = j / tpad; row = 2 + f * ZK_TOUT + t_real + (j - f * tpad); }
  const int c4 = threadIdx.x;
  const f4_t v = *(const f4_t*)(table + (size_t)row * ZK_HIDDEN + c4 * 4);
#pragma unroll
  for (int w = 0; w < 8; ++w)
    if (b0 + w < n_windows)
      __builtin_nontemporal_store(v, (f4_t*)(hidden + ((size_t)(b0 + w) * ZK_SEQ + row) * ZK_HIDDEN + c4 * 4));
}

// one workgroup = one token row of ZK_L0_WPB consecutive windows: 288 threads x 16 bytes, both planes.  Which source the row
// comes from is decided once per workgroup; a constant row is read from the table ONCE and stored ZK_L0_WPB times
#define ZK_L0_WPB 8
__global__ __launch_bounds__(320) void l0_assemble_qkv_kernel(const half_t* __restrict__ r_hi, const half_t* __restrict__ r_lo,
                                                              const half_t* __restrict__ t_hi, const half_t* __restrict__ t_lo,
                                                              half_t* __restrict__ o_hi, half_t* __restrict__ o_lo, int n_windows,
                                                              int t_real) {
  constexpr int CH = 3 * ZK_HIDDEN / 8, LD = 3 * ZK_HIDDEN;
  const int row = blockIdx.x % ZK_SEQ, b0 = (blockIdx.x / ZK_SEQ) * ZK_L0_WPB;
  const int ch = threadIdx.x;
  if (ch >= CH) return;
  int f = 0, t = ZK_TOUT;
  if (row >= 2) { const int p = row - 2; f = p / ZK_TOUT; t = p - f * ZK_TOUT; }
  const bool real = t < t_real;
  h8_t vh, vl;
  if (!real) {
    vh = *(const h8_t*)(t_hi + (size_t)row * LD + ch * 8);
    if (o_lo) vl = *(const h8_t*)(t_lo + (size_t)row * LD + ch * 8);
  }
#pragma unroll 4
  for (int w = 0; w < ZK_L0_WPB; ++w) {
    const int b = b0 + w;
    if (b >= n_windows) break;
    if (real) {
      const size_t r = ((size_t)b * ZK_FOUT + f) * t_real + t;
      vh = *(const h8_t*)(r_hi + r * LD + ch * 8);
      if (o_lo) vl = *(const h8_t*)(r_lo + r * LD + ch * 8);
    }
    const size_t oo = ((size_t)b * ZK_SEQ + row) * LD + ch * 8;
    // (non-temporal: 11 MB per window that the attention kernel reads back from HBM much later)
    __builtin_nontemporal_store(vh, (h8_t*)(o_hi + oo));
    if (o_lo) __builtin_nontemporal_store(vl, (h8_t*)(o_lo + oo));
  }
}

// ---- layer-0 constant-row ATTENTION (attention.hip: att_ext; zkast.hip) -------------------------------------------------
// constant-order table: row c of the output = q|k|v row of constant token c (0, 1, then (f, t >= t_real) ascending) of the
// natural-order table planes; rows n_const .. rows_pad-1 are zero (whole 64-row tiles, finite values behind masked keys)
__global__ __launch_bounds__(320) void l0_const_order_kernel(const half_t* __restrict__ t_hi, const half_t* __restrict__ t_lo,
                                                             half_t* __restrict__ o_hi, half_t* __restrict__ o_lo, int t_real, int n_const) {
  constexpr int CH = 3 * ZK_HIDDEN / 8, LD = 3 * ZK_HIDDEN;
  const int c = blockIdx.x, ch = threadIdx.x;
  if (ch >= CH) return;
  h8_t vh = {}, vl = {};
  if (c < n_const) {
    int row = c;
    if (c >= 2) { const int j = c - 2, tpad = ZK_TOUT - t_real, f = j / tpad; row = 2 + f * ZK_TOUT + t_real + (j - f * tpad); }
    vh = *(const h8_t*)(t_hi + (size_t)row * LD + ch * 8);
    if (t_lo) vl = *(const h8_t*)(t_lo + (size_t)row * LD + ch * 8);
  }
  *(h8_t*)(o_hi + (size_t)c * LD + ch * 8) = vh;
  if (o_lo) *(h8_t*)(o_lo + (size_t)c * LD + ch * 8) = vl;
}

// per-window TAIL planes [n_windows][128][2304]: rows 0 .. rem-1 = the last `rem` constant rows (rows 1088 .. of the
// constant-order table), rows rem .. rem + 12·t_real - 1 = the window's real rows (compact [b][f][t < t_real] planes of the
// layer-0 QKV GEMM), the rest zero.  One workgroup = one tail row of ZK_L0_WPB consecutive windows.
__global__ __launch_bounds__(320) void l0_tail_kernel(const half_t* __restrict__ r_hi, const half_t* __restrict__ r_lo,
                                                      const half_t* __restrict__ c_hi, const half_t* __restrict__ c_lo,
                                                      half_t* __restrict__ o_hi, half_t* __restrict__ o_lo, int n_windows, int t_real, int rem) {
  constexpr int CH = 3 * ZK_HIDDEN / 8, LD = 3 * ZK_HIDDEN;
  const int row = blockIdx.x % 128, b0 = (blockIdx.x / 128) * ZK_L0_WPB;
  const int ch = threadIdx.x;
  if (ch >= CH) return;
  const int n_real = ZK_FOUT * t_real;
  h8_t vh = {}, vl = {};
  if (row < rem) {
    vh = *(const h8_t*)(c_hi + (size_t)(1088 + row) * LD + ch * 8);
    if (o_lo) vl = *(const h8_t*)(c_lo + (size_t)(1088 + row) * LD + ch * 8);
  }
  const bool real = row >= rem && row < rem + n_real;
#pragma unroll 4
  for (int w = 0; w < ZK_L0_WPB; ++w) {
    const int b = b0 + w;
    if (b >= n_windows) break;
    if (real) {
      const size_t r = (size_t)b * n_real + (row - rem);
      vh = *(const h8_t*)(r_hi + r * LD + ch * 8);
      if (o_lo) vl = *(const h8_t*)(r_lo + r * LD + ch * 8);
    }
    const size_t oo = ((size_t)b * 128 + row) * LD + ch * 8;
    *(h8_t*)(o_hi + oo) = vh;
    if (o_lo) *(h8_t*)(o_lo + oo) = vl;
  }
}

}  // namespace

void zk_launch_l0_const_order(zk_planes table, zk_planes out, int t_real, int rows_pad, hipStream_t s) {
  const int n_const = 2 + ZK_FOUT * (ZK_TOUT - t_real);
  hipLaunchKernelGGL(l0_const_order_kernel, dim3((unsigned)rows_pad), dim3(320), 0, s, table.hi, table.lo, out.hi, out.lo, t_real, n_const);
}

void zk_launch_l0_tail(zk_planes real_rows, zk_planes ctab, zk_planes out, int n_windows, int t_real, hipStream_t s) {
  if (n_windows <= 0) return;
  const int rem = 2 + ZK_FOUT * (ZK_TOUT - t_real) - 1088;
  hipLaunchKernelGGL(l0_tail_kernel, dim3((unsigned)(((n_windows + ZK_L0_WPB - 1) / ZK_L0_WPB) * 128)), dim3(320), 0, s, real_rows.hi, real_rows.lo,
                     ctab.hi, ctab.lo, out.hi, out.lo, n_windows, t_real, rem);
}

void zk_launch_gather_tok01(zk_planes att, const float* hidden, int n_windows, zk_planes att_out, float* hidden_out,
                            hipStream_t s) {
  if (n_windows <= 0) return;
  const int total = n_windows * 2 * (ZK_HIDDEN / 4);
  hipLaunchKernelGGL(gather_tok01_kernel, dim3((total + 255) / 256), dim3(256), 0, s, att.hi, att.lo, hidden, n_windows,
                     att_out.hi, att_out.lo, hidden_out, att.tiled);
}

void zk_launch_im2col_compact(const float* feats, int n_frames, const int32_t* win_idx, int n_windows, float mean,
                              float std2, zk_planes out, hipStream_t s, int t_real) {
  if (n_windows <= 0) return;
  const int64_t total = (int64_t)n_windows * (t_real ? ZK_FOUT * t_real : ZK_NPATCH) * 32;
  hipLaunchKernelGGL(im2col_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, feats, n_frames,
                     win_idx, n_windows, mean, std2, out.hi, out.lo, out.lo_fmt, t_real);
}

void zk_launch_gather_xq(zk_planes x, int n_windows, zk_planes out, hipStream_t s) {
  if (n_windows <= 0) return;
  const int total = n_windows * ZK_QROWS * (ZK_HIDDEN / 8);
  hipLaunchKernelGGL(gather_xq_kernel, dim3((total + 255) / 256), dim3(256), 0, s, x.hi, x.lo, x.rowexp, x.tiled, n_windows, out.hi,
                     out.lo, out.rowexp);
}

void zk_launch_scatter_q(zk_planes q, int n_windows, zk_planes qkv, hipStream_t s) {
  if (n_windows <= 0) return;
  const int total = n_windows * ZK_QROWS * (ZK_HIDDEN / 8);
  hipLaunchKernelGGL(scatter_q_kernel, dim3((total + 255) / 256), dim3(256), 0, s, q.hi, q.lo, n_windows, qkv.hi, qkv.lo);
}

void zk_launch_l0_fill_hidden(float* hidden, const float* table, int n_windows, int t_real, hipStream_t s) {
  if (n_windows <= 0) return;
  const int nconst = 2 + ZK_FOUT * (ZK_TOUT - t_real);
  hipLaunchKernelGGL(l0_fill_hidden_kernel, dim3((unsigned)(((n_windows + 7) / 8) * nconst)), dim3(ZK_HIDDEN / 4), 0, s, hidden, table, n_windows, t_real);
}

void zk_launch_l0_assemble_qkv(zk_planes real_rows, zk_planes table, zk_planes out, int n_windows, int t_real, hipStream_t s) {
  if (n_windows <= 0) return;
  hipLaunchKernelGGL(l0_assemble_qkv_kernel, dim3((unsigned)(((n_windows + ZK_L0_WPB - 1) / ZK_L0_WPB) * ZK_SEQ)), dim3(320), 0, s, real_rows.hi, real_rows.lo,
                     table.hi, table.lo, out.hi, out.lo, n_windows, t_real);
}

void zk_launch_im2col_full(const float* input_values, int n_windows, zk_planes out, hipStream_t s) {
  if (n_windows <= 0) return;
  const int64_t total = (int64_t)n_windows * ZK_NPATCH * 32;
  hipLaunchKernelGGL(im2col_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, input_values,
                     ZK_MAXLEN, (const int32_t*)nullptr, n_windows, 0.f, 1.f, out.hi, out.lo, out.lo_fmt, 0);
}

void zk_launch_cls_rows(float* hidden, const float* cls, const float* dist, const float* pos, int n_windows,
                        hipStream_t s) {
  if (n_windows <= 0) return;
  const int total = n_windows * 2 * ZK_HIDDEN;
  hipLaunchKernelGGL(cls_rows_kernel, dim3((total + 255) / 256), dim3(256), 0, s, hidden, cls, dist, pos, n_windows);
}
