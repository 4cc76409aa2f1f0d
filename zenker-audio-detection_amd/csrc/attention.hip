// Flash-style multi-head self-attention for AST: S = 1214 tokens, 12 heads, d = 64, no mask.
// Replaces ASTAttention's softmax(q·kᵀ·d^-½)·v ($TF/.../modeling_audio_spectrogram_transformer.py:102-127,159-172);
// the 12×1214² score matrix (70.7 MB / window) never leaves registers.
//
// gfx950 design
//  * workgroup = 4 waves = 128 query rows of one (window, head); each wave owns 32 query rows.
//  * swapped product  Sᵀ[key][q] = K·Qᵀ  with v_mfma_f32_32x32x16_f16: the query index lands on the LANE and the
//    32 keys of a block in that lane's 16 registers (x2 half-waves), so the row max / row sum are in-lane
//    reductions plus ONE cross-half exchange, and the probability tile is, after an in-register cvt to fp16,
//    directly the B operand of  Oᵀ[d][q] += Vᵀ·Pᵀ  (no LDS round trip for P).
//  * K tile [64 keys][64 d] in LDS, 16-B chunk swizzle (key>>1)&7 (conflict-free ds_read_b128 for the 32x32x16
//    A-operand pattern); V tile [64 keys][64 d] read TRANSPOSED with ds_read_b64_tr_b16, swizzle ((key>>1)&1)<<2.
//  * K/V tiles are prefetched global->VGPR during the MFMA phase and written to the other LDS buffer after it.
//  * NSPLIT=3: q and k are (hi, lo) fp16 pairs, Sᵀ += Kh·Qh + Kl·Qh + Kh·Ql (the scores feed exp(), which
//    amplifies operand rounding); P·V stays single-pass (measured contribution 8e-5 on the logits).
//  * softmax in the log2 domain (v_exp_f32): the scale is folded into q, the negated running max is the C operand of
//    the score MFMAs (no per-score subtract), O/l are rescaled only when the max moves by more than 2^8 (deferred
//    rescale); fp32 running max / sum, keys >= 1214 of the last tile masked.
#include "zk_common.h"

namespace {

constexpr int S_ = ZK_SEQ;
constexpr int QKV_LD = 3 * ZK_HIDDEN;   // 2304
constexpr int KT = 64;                  // keys per tile
constexpr int NKT = (S_ + KT - 1) / KT; // 19
constexpr int QT = 128;                 // query rows per workgroup
constexpr int NQT = (S_ + QT - 1) / QT; // 10
constexpr int TILE_B = KT * 128;        // bytes of one [64][64] fp16 image

template <int NSPLIT>
__global__ __launch_bounds__(256) void attention_kernel(const half_t* __restrict__ qkv_hi,
                                                        const half_t* __restrict__ qkv_lo, half_t* __restrict__ o_hi,
                                                        half_t* __restrict__ o_lo, int n_windows, int q_tiles, int lo_fmt) {
  constexpr bool SPLIT = (NSPLIT == 3);
  constexpr int NIMG = SPLIT ? 3 : 2;     // Kh, [Kl], V
  constexpr int BUF_B = NIMG * TILE_B;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5;

  // XCD-aware bijective remap: the 10 query tiles of one (window, head) share K/V -> keep them on one XCD.
  const int nwg = q_tiles * ZK_HEADS * n_windows;
  int wg;
  {
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int qt = wg % q_tiles;
  const int head = (wg / q_tiles) % ZK_HEADS;
  const int win = wg / (q_tiles * ZK_HEADS);
  const size_t tok0 = (size_t)win * S_;

  // ---- Q fragments (B operand: lane holds Q[q = lane&31][d = 16ks + 8*half + j]) ----
  const int q_row = qt * QT + wave * 32 + (lane & 31);
  const int q_ld = q_row < S_ ? q_row : S_ - 1;
  h8_t qh[4], ql[SPLIT ? 4 : 1];
  {
    const size_t off = (tok0 + q_ld) * QKV_LD + head * ZK_HEAD_DIM + 8 * half;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qh[ks] = *(const h8_t*)(qkv_hi + off + ks * 16);
      if constexpr (SPLIT) ql[ks] = *(const h8_t*)(qkv_lo + off + ks * 16);
      // fold the softmax scale and the change to the log2 domain into q once: q <- q * (d^-1/2 * log2 e)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float qf = (float)qh[ks][e];
        if constexpr (SPLIT) qf += (float)ql[ks][e];
        qf *= 0.125f * 1.4426950408889634f;
        qh[ks][e] = (half_t)qf;
        if constexpr (SPLIT) ql[ks][e] = (half_t)(qf - (float)qh[ks][e]);
      }
    }
  }

  // ---- K/V staging (global -> VGPR -> LDS) ----
  // 512 16-B chunks per image; thread handles chunks tid and tid+256: row = c>>3, col chunk = c&7
  h8_t pk[NIMG][2];
  // per-lane byte offsets of this thread's two 16-B chunks inside a 64-key tile, computed once (the last tile clamps
  // keys >= 1214 to 1213): a tile load is then 6 x global_load_dwordx4 v, v_off, s[base] with no vector address math
  unsigned toff[2];
  auto set_offs = [&](int kt) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = tid + u * 256;
      const int row = c >> 3, cc = c & 7;
      int key = kt * KT + row;
      key = key < S_ ? key : S_ - 1;
      toff[u] = (unsigned)(key - kt * KT) * (unsigned)(QKV_LD * 2) + (unsigned)(cc * 16);
    }
  };
  set_offs(0);
  auto uniform_ptr = [](const char* p) {
    const unsigned long long g = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)g);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(g >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
  };
  // The loads name the GLOBAL address space: a pointer rebuilt from readfirstlane'd integers is otherwise generic and
  // hipcc emits flat_load, which also counts in lgkmcnt — every wait for a K/V fragment read of the CURRENT tile then
  // waits for the HBM round trip of the NEXT tile's prefetch.
  typedef __attribute__((address_space(1))) h8_t GLOBAL_H8;
  auto load_tile = [&](int kt) {
    if (kt == NKT - 1) set_offs(kt);
    const size_t tb = ((tok0 + (size_t)kt * KT) * QKV_LD + head * ZK_HEAD_DIM) * 2;     // bytes
    const char* gk = uniform_ptr((const char*)qkv_hi + tb + ZK_HIDDEN * 2);
    const char* gv = uniform_ptr((const char*)qkv_hi + tb + 2 * ZK_HIDDEN * 2);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      pk[0][u] = *(const GLOBAL_H8*)(gk + toff[u]);
      if constexpr (SPLIT) {
        const char* gl = uniform_ptr((const char*)qkv_lo + tb + ZK_HIDDEN * 2);
        pk[1][u] = *(const GLOBAL_H8*)(gl + toff[u]);
      }
      pk[NIMG - 1][u] = *(const GLOBAL_H8*)(gv + toff[u]);
    }
  };
  auto store_tile = [&](int buf) {
    char* base = smem + buf * BUF_B;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = tid + u * 256;
      const int row = c >> 3, cc = c & 7;
      const int kofs = row * 128 + ((cc ^ ((row >> 1) & 7)) << 4);
      *(h8_t*)(base + kofs) = pk[0][u];
      if constexpr (SPLIT) *(h8_t*)(base + TILE_B + kofs) = pk[1][u];
      const int vofs = row * 128 + ((cc ^ (((row >> 1) & 1) << 2)) << 4);
      *(h8_t*)(base + (NIMG - 1) * TILE_B + vofs) = pk[NIMG - 1][u];
    }
  };

  // ---- per-lane LDS read offsets ----
  const int kfrag_row = (lane & 31) * 128;
  const int kfrag_sw = (lane >> 1) & 7;
  // V transposed read: 16-lane group g: dblock = g&1, h = g>>1; lane i in group: key row i>>2, 4 d-columns at 4*(i&3)
  const int vg = lane >> 4, vi = lane & 15;
  const int v_key = 4 * (vg >> 1) + (vi >> 2);                 // + 32kb + 16s (+8)
  const int v_chunk = 2 * (vg & 1) + ((vi & 3) >> 1);          // + 4mb
  const int v_sw = ((v_key >> 1) & 1) << 2;
  const int v_byte = (vi & 1) * 8;

  f16_t oacc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { oacc[0][i] = 0.f; oacc[1][i] = 0.f; }
  // running max m_run (log2 domain) is kept NEGATED in a 16-register vector that is the C operand of each score
  // block's first MFMA: the accumulator then holds s - m_run directly and exp2 needs no subtraction.  m_run only
  // moves when some row's block maximum exceeds it by more than RESCALE_THR (p <= 2^8 is harmless in fp16/fp32);
  // that rare path rescales O and l (T13-style deferred rescale, wave-uniform branch).
  constexpr float RESCALE_THR = 8.0f;
  f16_t negm;
#pragma unroll
  for (int i = 0; i < 16; ++i) negm[i] = 0.f;
  float m_run = 0.f, l_run = 0.f;

  const bool wave_active = __builtin_amdgcn_readfirstlane(qt * QT + wave * 32) < S_;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < NKT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < NKT) load_tile(kt + 1);
    const char* kb_base = smem + cur * BUF_B;
    const char* vb_base = kb_base + (NIMG - 1) * TILE_B;

    // a wave whose 32 query rows all lie beyond the sequence (the last query tile holds 62 of 128 rows: waves 2 and 3)
    // only takes part in the K/V staging and the barriers — its SIMD time goes to the other resident workgroup
    if (wave_active) {
    // ---- scores minus running max: two 32-key blocks ----
    f16_t sacc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int ofs = kb * 32 * 128 + kfrag_row + (((2 * ks + half) ^ kfrag_sw) << 4);
        const h8_t kh = *(const h8_t*)(kb_base + ofs);
        if constexpr (SPLIT) {
          const h8_t kl = *(const h8_t*)(kb_base + TILE_B + ofs);
          sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], ks == 0 ? negm : sacc[kb], 0, 0, 0);
          sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], sacc[kb], 0, 0, 0);
          sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], sacc[kb], 0, 0, 0);
        } else {
          sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], ks == 0 ? negm : sacc[kb], 0, 0, 0);
        }
      }
    }

    // ---- online softmax (log2 domain); lane = query, registers = keys ----
    if (kt == NKT - 1) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * KT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (key >= S_) sacc[kb][r] = -1e30f;
        }
    }
    float mx = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(sacc[0][r], sacc[1][r]));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (kt == 0 || !__all(mx <= RESCALE_THR)) {
      const float delta = (kt == 0) ? mx : fmaxf(mx, 0.f);
      const float alpha = (kt == 0) ? 1.0f : __builtin_amdgcn_exp2f(-delta);
      m_run += delta;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sacc[0][r] -= delta; sacc[1][r] -= delta; }
#pragma unroll
      for (int i = 0; i < 16; ++i) { oacc[0][i] *= alpha; oacc[1][i] *= alpha; negm[i] = -m_run; }
      l_run *= alpha;
    }
    float psum = 0.f;
    h8_t pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(sacc[kb][r]);
        psum += p;
        pf[kb][r >> 3][r & 7] = (half_t)p;
      }
    l_run += psum;

    // ---- Oᵀ[d][q] += Vᵀ · Pᵀ ----
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          const int key0 = kb * 32 + 16 * s + v_key;
          const int a0 = key0 * 128 + (((4 * mb + v_chunk) ^ v_sw) << 4) + v_byte;
          const s4v_t t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s4v_t*)(vb_base + a0));
          const s4v_t t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s4v_t*)(vb_base + a0 + 8 * 128));
          const h4_t x0 = __builtin_bit_cast(h4_t, t0), x1 = __builtin_bit_cast(h4_t, t1);
          const h8_t vt = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
          oacc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vt, pf[kb][s], oacc[mb], 0, 0, 0);
        }
    }   // wave_active

    if (kt + 1 < NKT) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---- finalize: O / l, store 4 consecutive d per register group ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (q_row < S_) {
    const size_t obase = (tok0 + q_row) * ZK_HIDDEN + head * ZK_HEAD_DIM;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int d = 32 * mb + 8 * rg + 4 * half;
        h4_t hi;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = oacc[mb][4 * rg + j] * inv;
          hi[j] = (half_t)v[j];
        }
        *(h4_t*)(o_hi + obase + d) = hi;
        if (o_lo) *(h4_t*)(o_lo + obase + d) = zk_lo4(v, hi, lo_fmt);
      }
  }
}

}  // namespace

void zk_launch_attention(zk_planes qkv, zk_planes out, int n_windows, int nsplit, int q_tiles, hipStream_t s) {
  if (n_windows <= 0) return;
  if (q_tiles <= 0 || q_tiles > NQT) q_tiles = NQT;   // q_tiles < 10: only the first q_tiles*128 query rows (last-layer pruning)
  const int grid = q_tiles * ZK_HEADS * n_windows;
  if (nsplit == 3) {
    hipLaunchKernelGGL(attention_kernel<3>, dim3(grid), dim3(256), 2 * 3 * TILE_B, s, qkv.hi, qkv.lo, out.hi, out.lo,
                       n_windows, q_tiles, out.lo_fmt);
  } else {
    hipLaunchKernelGGL(attention_kernel<1>, dim3(grid), dim3(256), 2 * 2 * TILE_B, s, qkv.hi, qkv.lo, out.hi, out.lo,
                       n_windows, q_tiles, out.lo_fmt);
  }
}
