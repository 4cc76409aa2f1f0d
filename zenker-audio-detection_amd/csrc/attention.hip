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
//  * K/V tiles arrive by LDS-DMA (global_load_lds_dwordx4, swizzles applied on the source side) into 3-deep rings; the
//    loop is a sequence of fenced slots — one MFMA group, the fragment reads of the slot four ahead, a share of the
//    vector work — with K running one tile ahead of V (see "one key tile as a sequence of SLOTS" below).
//  * NSPLIT=3: q and k are (hi, lo) fp16 pairs, Sᵀ += Kh·Qh + Kl·Qh + Kh·Ql (the scores feed exp(), which
//    amplifies operand rounding); P·V stays single-pass (measured contribution 8e-5 on the logits).
//  * NSPLIT=2 (ZK_F16C8): the two correction products ride ONE fp8 pass as in gemm_c8.hip: k's lo plane holds the c8
//    byte pairs (fp8(kl·2^11), fp8(k)) written by the QKV epilogue, the kernel builds q' = (fp8(q), fp8(ql·2^11)) once
//    in registers, and Sᵀ += Kh·Qh (4 x 32x32x16 fp16) + K'·Q'·2^-11 (2 x v_mfma_scale_f32_32x32x64_f8f6f4) per 32
//    keys: 256 matrix-pipe cycles instead of 384.  The byte order inside a lane's 32 K'-bytes is the order of the
//    lane's fp16 fragments (d = 16ks + 8·half + j), so the c8 fragment reads use the Kh addresses in the next image.
//  * softmax in the log2 domain (v_exp_f32): the scale is folded into q, the negated running max is the C operand of
//    the score MFMAs (no per-score subtract), O/l are rescaled only when the max moves by more than 2^8 (deferred
//    rescale); fp32 running max / sum, keys >= 1214 of the last tile masked.
#include <type_traits>
#include <utility>

#include "zk_common.h"

namespace {

#ifndef ZK_ATT_DMA_TOP
#define ZK_ATT_DMA_TOP 0      // 1: all staging pieces at the top of the iteration instead of between the MFMA slots
#endif
#ifndef ZK_ATT_ABL
#define ZK_ATT_ABL 0      // probe builds only (timing, wrong results): 1 no K/V staging, 2 no barrier, 4 no exponentials, 8 half the staging, 16 no output stores
#endif
typedef int i2v_t __attribute__((ext_vector_type(2)));
typedef int i4v_t __attribute__((ext_vector_type(4)));
typedef int i8v_t __attribute__((ext_vector_type(8)));

// compile-time loop: f(integral_constant<int, I>) for I in [0, N)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

constexpr int S_ = ZK_SEQ;
constexpr int QKV_LD = 3 * ZK_HIDDEN;   // 2304
constexpr int KT = 64;                  // keys per tile
constexpr int NKT_FULL = (S_ + KT - 1) / KT; // 19
// ---- generalised launches (layer-0 constant-row attention, zkast.hip) ----
// VAR 0 is the ordinary launch: 1214 queries x 1214 keys of every window, rows in token order.  VAR >= 1 reads its key
// tiles from TWO sources — tiles [0, seg) from planes shared by all windows (source A: the layer-0 table in constant-token
// order), the others from the window's own planes (source B, b_rows rows per window) —, may take its query rows from A, masks
// keys >= n_keys, maps output row r to a token (out_map), and may start from (VAR 2) or dump (VAR 3) the running softmax
// state (O[64] unnormalised, m, l per head and query row).  All sources hold whole 64-row tiles (no clamping of the last one).
struct att_ext {
  const half_t* a_hi; const half_t* a_lo;
  float* state;        // [ZK_HEADS][state_rows][ST_LD] fp32
  int seg, a_row0;     // key tile kt < seg: rows a_row0 + kt*64 of A; kt >= seg: rows (kt - seg)*64 of the window's planes
  int a_q_row0;        // q_from_a: query row r is row a_q_row0 + r of A
  int b_rows;          // rows per window of the per-window planes (1214 for VAR 0)
  int n_keys;          // valid keys
  int n_q, q_lo;       // query rows per (window, head); rows q_lo <= r < n_q produce output
  int q_from_a;
  int out_map;         // 1: r = constant token r (0, 1, then (f, t >= tr) ascending); 2: r - q_lo = real token f*tr + t
  int tr;              // real time patches per frequency row (layer-0 reuse: 10)
  int state_rows;
};
constexpr int ST_LD = 68;      // floats per state row: O[0..63], m, l (+2 pad)
// token of output row r under out_map (tr real time patches per frequency row, ZK_TOUT = 101 in all)
__device__ __forceinline__ int att_out_token(int r, int out_map, int q_lo, int tr) {
  if (out_map == 1) { if (r < 2) return r; const int c = r - 2, nc = ZK_TOUT - tr, f = c / nc; return 2 + f * ZK_TOUT + tr + (c - f * nc); }
  if (out_map == 2) { const int j = r - q_lo, f = j / tr; return 2 + f * ZK_TOUT + (j - f * tr); }
  return r;
}
#ifndef ZK_ATT_NW
#define ZK_ATT_NW 8      // waves per workgroup: 256 query rows share a K/V tile; ONE workgroup per CU (the split modes' four
                         // images in 3-deep rings are 96 KiB), half the staging pieces per wave of the 4-wave form (same speed)
#endif
constexpr int NW = ZK_ATT_NW;
#ifndef ZK_ATT_STAGGER
#define ZK_ATT_STAGGER 1      // waves 4-7 run half an iteration behind their SIMD partners (see "staggered waves" below); 0 = lockstep form
#endif
#ifndef ZK_ATT_PERSIST_L0
#define ZK_ATT_PERSIST_L0 0     // probe: 2 = the layer-0 constant-query launch (att_ext, VAR 2) in the persistent form.  Its items are three
                                // key tiles long (launch + prologue + store tail are most of an item), but inside the item loop hipcc
                                // needs 256 VGPRs + 120-135 spilled registers for it (230 and none outside): not shipped
#endif
#ifndef ZK_ATT_PERSIST
#define ZK_ATT_PERSIST 0      // probe (measured +1.0 % in f16c8, -0.5 % in f16x3: profiles/r03_attention_stagger_ab.txt): 1 / 2 = one workgroup per
                              // CU walks its XCD's share of the (window, head, query tile) items (2: q loads behind the stores, no spills)
#endif
constexpr int NVS = ZK_ATT_STAGGER ? 4 : 3;      // V ring slots (the late waves read V(t-1) while V(t+2) is being staged)
constexpr int QT = 32 * NW;             // query rows per workgroup
constexpr int TILE_B = KT * 128;        // bytes of one [64][64] fp16 image

template <int NSPLIT, int NKT, int VAR>
__global__ __launch_bounds__(64 * NW) void attention_kernel(const half_t* __restrict__ qkv_hi,
                                                        const half_t* __restrict__ qkv_lo, half_t* __restrict__ o_hi,
                                                        half_t* __restrict__ o_lo, int n_windows, int q_tiles, int lo_fmt, int row_limit, int rev,
                                                        int o_tiled, const att_ext x) {
  constexpr bool GEN = VAR != 0, INIT = VAR == 2, DUMP = VAR == 3;
  // persistent form (one workgroup per CU walks its XCD's share of the items): probe switches for the ordinary launch
  // (ZK_ATT_PERSIST) and for the layer-0 constant-query launch (VAR 2, ZK_ATT_PERSIST_L0)
  constexpr int PERS = VAR == 2 ? ZK_ATT_PERSIST_L0 : (GEN ? 0 : ZK_ATT_PERSIST);
  constexpr bool SPLIT = (NSPLIT >= 2);
  constexpr bool C8 = (NSPLIT == 2);
  // LDS images of one 64-key tile: K: Kh, [Kl | Kc8]; V: Vh, [Vl].  Vl = fp16(v - fp16(v)), the lo plane the QKV epilogue
  // writes for v's columns: P·V = P·Vh + P·Vl.  v used to be the ONE operand of the whole network that reached an MFMA
  // with plain fp16 rounding (2^-11; everything else carries a correction to ~2^-15+): harmless while attention averages
  // over many keys, but on sharply peaked rows the output IS one v row and its rounding error goes straight into the
  // residual stream (input-sensitive weight set: 1.2e-3 logit error, 0.4e-3 with v corrected).  The second PV pass adds no
  // vector-ALU work (P is already there), only 8 MFMAs per tile on a pipe that was < 50 % busy.
#ifdef ZK_ATT_NO_VL      // probe builds only: no Vl pass (timing / accuracy A-B)
  constexpr bool VL = false;
#else
  constexpr bool VL = SPLIT;
#endif
  constexpr int NKIMG = SPLIT ? 2 : 1, NVIMG = VL ? 2 : 1, NIMG = NKIMG + NVIMG;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5;

  // XCD-aware bijective remap: the query tiles of one (window, head) share K/V -> keep them on one XCD.  XCD x (= block
  // index mod 8 under round-robin placement: speed only) owns a contiguous share of the nwg items.
  const int nwg = q_tiles * ZK_HEADS * n_windows;
  int qt, head;
  size_t tok0;      // first token row of the window in the OUTPUT planes (and, VAR 0, in the q|k|v planes)
  size_t brow0 = 0; // GEN: first row of the window in its own q|k|v planes
  const int n_q = GEN ? x.n_q : S_;      // query rows per (window, head)
  auto set_item = [&](int wg) __attribute__((always_inline)) {
    qt = wg % q_tiles;
    head = (wg / q_tiles) % ZK_HEADS;
    tok0 = (size_t)(wg / (q_tiles * ZK_HEADS)) * S_;
    if constexpr (GEN) brow0 = (size_t)(wg / (q_tiles * ZK_HEADS)) * x.b_rows;
  };
  // Persistent form: the workgroup walks items j, j + per, j + 2 per, ... of its XCD's share (at any moment the CUs of an
  // XCD work on neighbouring items, as the hardware's own dispatch order had it).  The NEXT item's q loads and K/V
  // prologue pieces are issued in front of THIS item's output stores, so the memory round trip of a prologue, the store
  // tail and the workgroup launch no longer sit exposed on a CU that holds only this one workgroup.
  int it_idx = 0, x_count = 0, x_per = 0, x_start = 0;
  auto item_wg = [&](int idx) __attribute__((always_inline)) { return x_start + (rev ? x_count - 1 - idx : idx); };      // rev: last item first
  if constexpr (PERS) {
    const int x_q = nwg >> 3, x_r = nwg & 7, x_id = (int)blockIdx.x & 7;
    x_per = (int)gridDim.x >> 3;
    x_start = x_id < x_r ? x_id * (x_q + 1) : x_r * (x_q + 1) + (x_id - x_r) * x_q;
    x_count = x_q + (x_id < x_r ? 1 : 0);
    it_idx = (int)blockIdx.x >> 3;
    if (it_idx >= x_count) return;
    set_item(item_wg(it_idx));
  } else {
    const int bid = rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;      // rev: last window first
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    set_item((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx);
  }

  // ---- Q fragments (B operand: lane holds Q[q = lane&31][d = 16ks + 8*half + j]) ----
  h8_t qh[4], ql[(SPLIT && !C8) ? 4 : 1];      // ql: q's fp16 lo fragments, kept only where the 3-term split reads them later
  i8v_t qc[C8 ? 2 : 1];      // C8: q' for the two 64-byte-deep fp8 MFMAs (ks = 2t, 2t+1)
  // The eight 16-byte loads are issued here, all at once; their scaling / re-splitting happens behind the issue of the
  // prologue's K/V staging pieces (q_prepare below), so that ONE memory round trip covers q, K(0..2) and V(0..1) — the
  // earlier order (load q, convert, then stage) paid three in a row per workgroup, with nothing else resident on the CU.
  h8_t qraw_h[4], qraw_l[SPLIT ? 4 : 1];
  auto q_issue = [&]() __attribute__((always_inline)) {      // (of the item set_item() selected)
    int ln;      // (read here, not taken from the kernel's `lane`: see the output path)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const int q_row = qt * QT + wave * 32 + (ln & 31);
    const int q_ld = q_row < n_q ? q_row : n_q - 1;
    size_t off = (tok0 + q_ld) * QKV_LD + head * ZK_HEAD_DIM + 8 * (ln >> 5);
    const half_t *qsh = qkv_hi, *qsl = qkv_lo;
    if constexpr (GEN) {
      off = ((x.q_from_a ? (size_t)x.a_q_row0 : brow0) + q_ld) * QKV_LD + head * ZK_HEAD_DIM + 8 * (ln >> 5);
      if (x.q_from_a) { qsh = x.a_hi; qsl = x.a_lo; }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qraw_h[ks] = *(const h8_t*)(qsh + off + ks * 16);
      if constexpr (SPLIT) qraw_l[ks] = *(const h8_t*)(qsl + off + ks * 16);
    }
  };
  q_issue();
  auto q_prepare = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qh[ks] = qraw_h[ks];
      // q's lo fragment of this k-step lives in a local: C8 needs it only here (ql[] then has ONE element, and indexing
      // it with ks would run past its end), the 3-term split keeps the re-split value in ql[ks] below
      h8_t qlo = {};
      if constexpr (SPLIT) qlo = qraw_l[ks];
      // fold the softmax scale and the change to the log2 domain into q once: q <- q * (d^-1/2 * log2 e)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float qf = (float)qh[ks][e];
        if constexpr (SPLIT) qf += (float)qlo[e];
        qf *= 0.125f * 1.4426950408889634f;
        qh[ks][e] = (half_t)qf;
        if constexpr (SPLIT && !C8) ql[ks][e] = (half_t)(qf - (float)qh[ks][e]);
        if constexpr (C8) {
          const int di = (ks & 1) * 4 + (e >> 1);
          const float l11 = zk_clamp_fp8((qf - (float)qh[ks][e]) * 2048.f);
          qc[ks >> 1][di] = (e & 1) ? __builtin_amdgcn_cvt_pk_fp8_f32(zk_clamp_fp8(qf), l11, qc[ks >> 1][di], true)
                                    : __builtin_amdgcn_cvt_pk_fp8_f32(zk_clamp_fp8(qf), l11, 0, false);
        }
      }
    }
  };

  // ---- K/V staging: LDS-DMA (global_load_lds_dwordx4) into 3-deep rings ----
  // The loop is software-pipelined by one tile: iteration t multiplies K(t+1)·Qᵀ BESIDE the softmax of tile t (the
  // matrix pipe holds the SIMD's issue for 8 of an MFMA's 32 cycles, the exp / max / cvt stream of the previous tile
  // fills the other 24), then V(t)ᵀ·Pᵀ.  So K runs one tile ahead of V.  Iteration t issues the DMA of K(t+3) and
  // V(t+2) into the ring slots that K(t) / V(t-1) left at the previous barrier and ends on vmcnt(PER_ITER): everything
  // but its own pieces has landed, i.e. a piece has a whole iteration (~1 us) to arrive, costs no VGPR and no ds_write.
  // A piece = one wave instruction = 8 rows x 128 B (1 KiB) of one image, lane -> (row = lane>>3, 16-B LDS chunk =
  // lane&7); the XOR swizzles of the images are applied on the SOURCE side (the DMA writes LDS lane-linearly).
  // A 64-key image has 8 pieces: wave w issues pieces w and w+4 (rows +32: same swizzle term).
  constexpr int PPI = 8 / NW;             // pieces per image and wave: row blocks wave, wave + NW, ...
  constexpr int PER_ITER = PPI * NIMG;    // DMA instructions per wave and iteration
  unsigned koff[PPI], voff[PPI], koff_last[PPI], voff_last[PPI];
#pragma unroll
  for (int u = 0; u < PPI; ++u) {
    const int row = (wave + NW * u) * 8 + (lane >> 3), cl = lane & 7;
    const unsigned kc = (unsigned)((cl ^ ((row >> 1) & 7)) << 4), vc = (unsigned)((cl ^ (((row >> 1) & 1) << 2)) << 4);
    koff[u] = (unsigned)row * (unsigned)(QKV_LD * 2) + kc;
    voff[u] = (unsigned)row * (unsigned)(QKV_LD * 2) + vc;
    int key = (NKT - 1) * KT + row;      // last tile: keys >= 1214 (masked later) re-read key 1213
    if constexpr (!GEN) key = key < S_ ? key : S_ - 1;      // (GEN: the sources hold whole tiles)
    const unsigned rl = (unsigned)(key - (NKT - 1) * KT) * (unsigned)(QKV_LD * 2);
    koff_last[u] = rl + kc;
    voff_last[u] = rl + vc;
  }
  auto uniform_ptr = [](const char* p) {
    const unsigned long long g = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)g);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(g >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
  };
  // LDS: K ring = 3 x NKIMG images at 0, V ring = NVS x NVIMG images behind it
  constexpr int KBUF_B = NKIMG * TILE_B, VBUF_B = NVIMG * TILE_B;
  constexpr int V_OFF = 3 * KBUF_B;
  auto dma = [&](const char* gbase, unsigned off, char* lds) __attribute__((always_inline)) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + off),
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
  };
  // piece pc (0 .. PER_ITER-1) of the iteration's staging: K images of tile kt_k -> K slot sk (pieces 0 .. PPI·NKIMG-1),
  // V images of tile kt_v -> V slot sv.  Tiles are clamped to the last one: the surplus fetches of the final iterations
  // land in dead slots.
  auto dma_piece = [&](int pc, int kt_k, int sk, int kt_v, int sv) __attribute__((always_inline)) {
    const bool isk = pc < PPI * NKIMG;
    int kt = isk ? kt_k : kt_v;
    kt = kt < NKT - 1 ? kt : NKT - 1;
    const bool last = kt == NKT - 1;
    const int u = pc % PPI, img = isk ? pc / PPI : pc / PPI - NKIMG;      // image 0 = hi plane, 1 = lo plane
    size_t tb = ((tok0 + (size_t)kt * KT) * QKV_LD + head * ZK_HEAD_DIM) * 2;     // bytes
    const char* src = (img == 1 ? (const char*)qkv_lo : (const char*)qkv_hi);
    if constexpr (GEN) {
      const bool from_a = kt < x.seg;
      tb = (((from_a ? (size_t)x.a_row0 + (size_t)kt * KT : brow0 + (size_t)(kt - x.seg) * KT)) * QKV_LD + head * ZK_HEAD_DIM) * 2;
      if (from_a) src = (img == 1 ? (const char*)x.a_lo : (const char*)x.a_hi);
    }
    src += tb + (isk ? 1 : 2) * ZK_HIDDEN * 2;
    char* base = isk ? smem + sk * KBUF_B + img * TILE_B : smem + V_OFF + sv * VBUF_B + img * TILE_B;
    const unsigned o = isk ? (last ? koff_last[u] : koff[u]) : (last ? voff_last[u] : voff[u]);
    dma(uniform_ptr(src), o, base + (wave + NW * u) * 1024);
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
  // moves when some row's block maximum exceeds it by more than RESCALE_THR; that path rescales O and l — and the
  // scores of the NEXT tile, which were started with the old maximum (T13-style deferred rescale, wave-uniform branch).
  // The threshold is a PRECISION knob, not only a range guard: the weights go to the P·V MFMA as fp16, and a weight
  // 2^(s - m_run) is exact exactly when s = m_run.  With the reference far below the row's true maximum (2^8 was
  // harmless for the range) the DOMINANT key of a peaked row carries a 2^-11 rounding error, which on the
  // input-sensitive weight set was the largest single term of the logit error (round 4, tools/sens_budget.py: 7.2e-4 at
  // 2^8, 2.7e-4 at 2^2 or below, where the rescale fires in 9 % of the wave-tiles).  At 2 a key that tops the reference by
  // more than 4x becomes the new (exact) reference; anything it leaves inexact shares its row with a comparable weight.
#ifndef ZK_ATT_RESCALE_THR
#define ZK_ATT_RESCALE_THR 2.0f
#endif
  constexpr float RESCALE_THR = ZK_ATT_RESCALE_THR;
  f16_t negm;
#pragma unroll
  for (int i = 0; i < 16; ++i) negm[i] = 0.f;
  float m_run = 0.f, l_run = 0.f;

  // ---- one key tile as a sequence of SLOTS ----
  // A slot = the LDS fragment reads of the slot LA (= 4) ahead, one group of MFMAs, and a share of the VALU work, fenced by
  // sched_barrier so that hipcc keeps the interleave (left alone it runs the MFMAs back to back, each behind its own
  // LDS round trip, and the softmax after them).
  //   score slots (tile t+1):  C8: per 32-key block 2 fp8 groups (1 MFMA, 64 cycles, 4 exps) + 4 fp16 groups (1 MFMA,
  //                            32 cycles, 2 exps); X3: 8 groups of 3 MFMAs, 4 exps; F16: 8 groups of 1 MFMA, 4 exps
  //   PV slots (tile t):       8 groups (split modes: 16, Vh and Vl alternating) of 1 MFMA + 4 (2) elements of the next
  //                            tile's row maximum
  struct kf_t { i4v_t a, b; };
  constexpr int NG_QK = C8 ? 12 : 8;
  constexpr int NG_PV = VL ? 16 : 8;      // split modes: every V fragment twice, from the Vh and the Vl image
  // (slots of a full sequence: NG_QK + NG_PV; of a Vh-only one NG_QK + NG_PV / 2 — see npv below)
#ifndef ZK_ATT_LA
#define ZK_ATT_LA 4
#endif
  constexpr int LA = ZK_ATT_LA;      // fragment reads run this many slots ahead (LA + 1 fragment register sets)
  kf_t fr[LA + 1];
  // Fragment reads are inline asm with hand-counted s_waitcnt lgkmcnt: hipcc cannot tell the LDS-DMA pieces of the ring
  // from the fragment reads and would put vmcnt(0) between them, and its own lgkmcnt placement drains the lookahead.
  // LDS reads retire in order, so the fragment of group g is valid once all but the reads issued after it have
  // returned: lgkmcnt(reads of groups g+1 .. g+LA).  Addresses are per-lane bases (+ ring slot, once per iteration)
  // plus immediates.
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  unsigned kofs[4], vofs[2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kofs[ks] = lds0 + (unsigned)(kfrag_row + (((2 * ks + half) ^ kfrag_sw) << 4));
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
    vofs[mb] = lds0 + (unsigned)(V_OFF + v_key * 128 + (((4 * mb + v_chunk) ^ v_sw) << 4) + v_byte);
  unsigned ka[4], va[2];      // the same with the ring slots of the iteration added
  auto nreads = [](int g) constexpr { return g >= NG_QK ? 2 : (C8 ? ((g >> 1) < 2 ? 2 : 1) : (SPLIT ? 2 : 1)); };
  static_assert(LA * 2 + 2 <= 15, "lgkmcnt is a 4-bit counter");
  // ---- the Vl·P pass only where it can matter ----
  // v's lo plane corrects the fp16 rounding of v (2^-11 relative).  A key whose softmax weight is below 2^-VL_SKIP_LOG2 of the
  // row's sum adds at most 2^-(11 + VL_SKIP_LOG2) of |v| through it, and keys that small come in crowds whose lo parts
  // average out (64 of them: 2^-15 of |v|, the size of the fp8 correction residues of the GEMMs).  So a wave runs the Vl·P
  // MFMAs of a key tile only if SOME row of its 32 has a key in the tile that weighs more than that against the row's sum so
  // far (skip_vl: a scalar branch around each Vl·P MFMA, mfma_vl below).  Their fragment reads stay: the counted waits of the
  // slot sequence depend on them, and a second, Vh-only instantiation of the sequence (vlt = false below: built, 2x the loop
  // code) sent hipcc's register allocation over the edge (256 VGPRs and > 1000 spills against 222 and none).  Measured on the
  // input-sensitive weight set (CPU emulation of the rule, tools/vl_skip_emul.py): 64 % of the (wave, tile) pairs skip at 2^-5,
  // logit error 1e-5 against 1.5e-3 for dropping the pass altogether; `wide` 75 %, `heavy` 87 %.
#ifndef ZK_ATT_VL_SKIP_LOG2
#define ZK_ATT_VL_SKIP_LOG2 5      // 0 = never skip (every tile runs the full sequence)
#endif
  // (not in the layer-0 launches, GEN: on the input-sensitive weight set a perturbation of layer 0 reaches the logits ~15x
  // amplified, that of any later layer hardly at all; the two layer-0 launches are 1 % of the attention time)
  constexpr bool VL_SKIP = VL && ZK_ATT_VL_SKIP_LOG2 > 0 && !GEN;
  constexpr float VL_TAU = VL_SKIP ? 1.0f / (float)(1 << ZK_ATT_VL_SKIP_LOG2) : 0.f;
  // One Vl·P MFMA behind a SCALAR branch, as ONE asm statement: a C++ `if` around the builtin splits the slot sequence into
  // ~40 basic blocks per tile, hipcc's allocator then needs 256 registers + spills where the straight-line form takes 222 — and
  // it spills fragment registers whose ds_read is still in flight (it cannot know the asm reads above are asynchronous:
  // NaNs).  Inside the asm nothing is visible to the allocator.  Hazards the compiler would have covered: 2 wait states
  // between a VALU write of P and the MFMA; `last`: the wait states before a VALU instruction (deferred rescale, finalize)
  // may read the accumulator (8-pass MFMA).
  auto mfma_vl = [](f16_t& acc, h8_t a, h8_t b, int skip, auto last_c) __attribute__((always_inline)) {
    const int sk = __builtin_amdgcn_readfirstlane(skip);      // ("s" alone does not move a value the allocator keeps in a VGPR)
    if constexpr (decltype(last_c)::value)
      asm volatile("s_cmp_lg_u32 %3, 0\n\ts_cbranch_scc1 .Lzk_vlskip%=\n\ts_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\t"
                   "s_nop 7\n\ts_nop 7\n\ts_nop 1\n.Lzk_vlskip%=:" : "+v"(acc) : "v"(a), "v"(b), "s"(sk) : "scc");
    else
      asm volatile("s_cmp_lg_u32 %3, 0\n\ts_cbranch_scc1 .Lzk_vlskip%=\n\ts_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n"
                   ".Lzk_vlskip%=:" : "+v"(acc) : "v"(a), "v"(b), "s"(sk) : "scc");
  };
  // position p of the early waves' slot order -> group id (the ids keep the full numbering: PV group NG_QK + 2v + img)
  auto npv = [](bool vlt) constexpr { return (VL && !vlt) ? NG_PV / 2 : NG_PV; };
  auto seq_g = [](int p, bool vlt) constexpr { return p < NG_QK ? p : NG_QK + ((VL && !vlt) ? 2 * (p - NG_QK) : p - NG_QK); };
  auto seq_reads_after = [nreads, seq_g](int p, int n_pos, bool vlt) constexpr {
    int n = 0;
    for (int j = p + 1; j <= p + LA && j < n_pos; ++j) n += nreads(seq_g(j, vlt));
    return n;
  };
  // (fic: which fragment register set — the position of the group in the wave's slot order modulo LA + 1)
  auto load_group_at = [&](auto gc, auto fic) __attribute__((always_inline)) {
    constexpr int g = decltype(gc)::value;
    kf_t& f = fr[decltype(fic)::value];
    if constexpr (g < NG_QK) {
      if constexpr (C8) {
        constexpr int kb = g & 1, i = g >> 1;      // the two 32-key blocks alternate: two independent accumulator chains
        if constexpr (i < 2) {      // c8 image, chunks of ks = 2i and 2i+1
          asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b128 %1, %3 offset:%4"
                       : "=&v"(f.a), "=&v"(f.b) : "v"(ka[2 * i]), "v"(ka[2 * i + 1]), "n"(kb * 4096 + TILE_B));
        } else {
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(f.a) : "v"(ka[i - 2]), "n"(kb * 4096));
        }
      } else {
        constexpr int kb = g & 1, ks = g >> 1;
        if constexpr (SPLIT)
          asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                       : "=&v"(f.a), "=&v"(f.b) : "v"(ka[ks]), "n"(kb * 4096), "n"(kb * 4096 + TILE_B));
        else
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(f.a) : "v"(ka[ks]), "n"(kb * 4096));
      }
    } else {
      // PV group: fragment (kb, sx, mb) of the Vh image — split modes: groups alternate Vh / Vl (image offset TILE_B)
      constexpr int pv = g - NG_QK, v = VL ? pv / 2 : pv, img = VL ? pv % 2 : 0;
      constexpr int kb = v / 4, sx = (v / 2) % 2, mb = v % 2;
      i2v_t t0, t1;
      asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                   : "=&v"(t0), "=&v"(t1) : "v"(va[mb]), "n"((kb * 32 + 16 * sx) * 128 + img * TILE_B),
                     "n"((kb * 32 + 16 * sx + 8) * 128 + img * TILE_B));
      f.a = __builtin_shufflevector(t0, t1, 0, 1, 2, 3);
    }
  };
  auto load_group = [&](auto gc) __attribute__((always_inline)) {
    load_group_at(gc, std::integral_constant<int, decltype(gc)::value % (LA + 1)>{});
  };
  // the fragment of group g is in its registers (see above); names them so that no MFMA moves in front of the wait
  auto wait_group_at = [&](auto gc, auto fic, auto n_c) __attribute__((always_inline)) {
    constexpr int g = decltype(gc)::value;
    kf_t& f = fr[decltype(fic)::value];
    if constexpr (nreads(g) == 2 && g < NG_QK)
      asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.a), "+v"(f.b) : "n"(decltype(n_c)::value));
    else
      asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f.a) : "n"(decltype(n_c)::value));
  };
  auto wait_group = [&](auto gc, auto n_c) __attribute__((always_inline)) {
    wait_group_at(gc, std::integral_constant<int, decltype(gc)::value % (LA + 1)>{}, n_c);
  };
  // the MFMAs of score group g into sn (scores of the next tile minus the running max)
  auto mma_group_at = [&](auto gc, auto fic, f16_t (&sn)[2]) __attribute__((always_inline)) {
    constexpr int g = decltype(gc)::value;
    const kf_t& f = fr[decltype(fic)::value];
    if constexpr (C8) {
      constexpr int kb = g & 1, i = g >> 1;      // the two 32-key blocks alternate: two independent accumulator chains
      if constexpr (i < 2) {
        const i8v_t kc = __builtin_shufflevector(f.a, f.b, 0, 1, 2, 3, 4, 5, 6, 7);
        sn[kb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kc, qc[i], i == 0 ? negm : sn[kb], 0, 0, 0,
                                                                 127 - ZK_C8_SHIFT, 0, 127);
      } else {
        sn[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_t, f.a), qh[i - 2], sn[kb], 0, 0, 0);
      }
    } else {
      constexpr int kb = g & 1, ks = g >> 1;
      const h8_t kh = __builtin_bit_cast(h8_t, f.a);
      if constexpr (SPLIT) {
        const h8_t kl = __builtin_bit_cast(h8_t, f.b);
        sn[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], ks == 0 ? negm : sn[kb], 0, 0, 0);
        sn[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], sn[kb], 0, 0, 0);
        sn[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], sn[kb], 0, 0, 0);
      } else {
        sn[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], ks == 0 ? negm : sn[kb], 0, 0, 0);
      }
    }
  };
  auto mma_group = [&](auto gc, f16_t (&sn)[2]) __attribute__((always_inline)) {
    mma_group_at(gc, std::integral_constant<int, decltype(gc)::value % (LA + 1)>{}, sn);
  };
  // first and count of the 32 exponentials that ride in score slot g
  auto exp_count = [](int g) constexpr { return C8 ? ((g >> 1) < 2 ? 4 : 2) : 4; };
  auto exp_first = [exp_count](int g) constexpr {
    int n = 0;
    for (int j = 0; j < g; ++j) n += exp_count(j);
    return n;
  };

  // a wave whose 32 query rows all lie beyond the sequence (the last query tile holds 62 of 128 rows: waves 2 and 3)
  // only takes part in the K/V staging and the barriers — its SIMD time goes to the other resident workgroup
  bool wave_active;
  auto kv_prologue = [&]() __attribute__((always_inline)) {      // K(0..2), V(0..1) of the item set_item() selected
#pragma unroll
    for (int pc = 0; pc < PER_ITER; ++pc) {
      dma_piece(pc, 0, 0, 0, 0);
      dma_piece(pc, 1, 1, 1, 1);
      if (pc < PPI * NKIMG) dma_piece(pc, 2, 2, 0, 0);
    }
  };
  kv_prologue();
  for (;;) {      // one item per pass (ZK_ATT_PERSIST = 0: a single pass)
  wave_active = __builtin_amdgcn_readfirstlane(qt * QT + wave * 32) < (row_limit < n_q ? row_limit : n_q);
#pragma unroll
  for (int i = 0; i < 16; ++i) { oacc[0][i] = 0.f; oacc[1][i] = 0.f; negm[i] = 0.f; }
  m_run = 0.f; l_run = 0.f;
  __builtin_amdgcn_sched_barrier(0);      // (the pieces are in flight before q's vector work starts)
  q_prepare();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if constexpr (INIT) {      // start from the tabulated state of this (head, query row): O (unnormalised), m, l over the keys it covers
    if (wave_active) {
      int ln;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
      int row = qt * QT + wave * 32 + (ln & 31);
      row = row < n_q ? row : n_q - 1;
      const float* st = x.state + ((size_t)head * x.state_rows + row) * ST_LD;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const f4_t v = *(const f4_t*)(st + 32 * mb + 8 * rg + 4 * (ln >> 5));
#pragma unroll
          for (int j = 0; j < 4; ++j) oacc[mb][4 * rg + j] = v[j];
        }
      m_run = st[64];
      l_run = (ln >> 5) == 0 ? st[65] : 0.f;      // (l_run is a per-half partial sum; the table holds the row's total)
#pragma unroll
      for (int i = 0; i < 16; ++i) negm[i] = -m_run;
    }
  }

  // row maximum of a score tile (in-lane over the 32 keys, then across the two half-waves)
  auto row_max = [&](f16_t (&sc)[2]) __attribute__((always_inline)) {
    float mx = fmaxf(sc[0][0], sc[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(sc[0][r], sc[1][r]));
    return fmaxf(mx, __shfl_xor(mx, 32, 64));
  };
  // move the running max by delta: scores of the current tile, O and l follow
  auto rescale = [&](f16_t (&sc)[2], float delta, float alpha) __attribute__((always_inline)) {
    m_run += delta;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sc[0][r] -= delta; sc[1][r] -= delta; }
#pragma unroll
    for (int i = 0; i < 16; ++i) { oacc[0][i] *= alpha; oacc[1][i] *= alpha; negm[i] = -m_run; }
    l_run *= alpha;
  };

  // Two probabilities p = 2^score of this lane's row: rounded to fp16 for the P·V MFMA, and the row sum l accumulates
  // the ROUNDED values, not the fp32 ones.  O / l is then an exactly normalised average with
  // weights fp16(p): a weight's rounding error enters as δ·p·(v − O) instead of δ·p·v, which vanishes for the sharply
  // peaked rows where one key carries the sum (measured on the input-sensitive weight set: logit error -28 %).
  // Layer-0 launches (GEN): the softmax weights go to the P·V MFMAs as (hi, lo) fp16 PAIRS — P·V = Ph·Vh + Pl·Vh + Ph·Vl, the
  // row sum over ph + pl.  On the input-sensitive weight set the fp16 rounding of P in layer 0 alone is the largest term left
  // once that layer's GEMMs run f16x3 (1.5e-4 rms of a logit, 70 % of what P's rounding costs over all twelve layers:
  // tools/sens_budget.py PER_LAYER), and the layer-0 launches are 1 % of the attention time: 8 more MFMAs per tile there, no
  // fragment read more (Pl·Vh rides in the Vh slot).
  constexpr bool PSPLIT = GEN && SPLIT;
  auto exp_pair = [&](f16_t (&sc)[2], h8_t (&pf)[2][2], h8_t (&pl)[2][2], int e, float acc) __attribute__((always_inline)) {
#if ZK_ATT_ABL & 4
    const float p0 = sc[e >> 4][e & 15], p1 = sc[e >> 4][(e & 15) + 1];
#else
    const float p0 = __builtin_amdgcn_exp2f(sc[e >> 4][e & 15]), p1 = __builtin_amdgcn_exp2f(sc[e >> 4][(e & 15) + 1]);
#endif
    const h2_t pr = {(half_t)p0, (half_t)p1};
    pf[e >> 4][(e & 15) >> 3][e & 7] = pr[0];
    pf[e >> 4][(e & 15) >> 3][(e & 7) + 1] = pr[1];
    // acc += fp16 halves of pr, exactly, as two v_fma_mix_f32 (fp16 source operands read in place: the same instruction
    // count as the fp32 adds they replace; v_dot2_f32_f16 would be one instruction but measured +5 % on the kernel)
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pr), "v"(acc));
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(acc) : "v"(pr), "v"(r));
    if constexpr (PSPLIT) {      // lo = fp16(p - fp16(p)) by one v_fma_mixlo / _mixhi each (zk_lo4's form), summed into l as well
      unsigned d;
      asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(pr), "v"(p0));
      asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(pr), "v"(p1));
      const h2_t lo = __builtin_bit_cast(h2_t, d);
      pl[e >> 4][(e & 15) >> 3][e & 7] = lo[0];
      pl[e >> 4][(e & 15) >> 3][(e & 7) + 1] = lo[1];
      asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(lo), "v"(acc));
      asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(acc) : "v"(lo), "v"(r));
    }
    return acc;
  };

  f16_t sA[2], sB[2];
  float mx = 0.f;
  // (the skip flags are wave-uniform 0 / 1 integers, forced into SGPRs by readfirstlane: mfma_vl compares them with s_cmp)
  auto all_small = [&](float mxv) __attribute__((always_inline)) {      // no key of the tile weighs more than VL_TAU of the row's sum so far
    return __builtin_amdgcn_readfirstlane(__all(__builtin_amdgcn_exp2f(mxv) <= VL_TAU * l_run) ? 1 : 0);
  };
  int skip_t0 = 0;      // tile 0 needs no Vl·P (possible only when a tabulated state precedes it: l_run > 0)
  int skip_vl = 0;      // the Vl·P MFMAs of the PV slots that run next are skipped
  if (wave_active) {      // scores of tile 0 (no overlap partner yet)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ka[ks] = kofs[ks];      // K(0) is in slot 0
    static_for<NG_QK>([&](auto gc) __attribute__((always_inline)) {
      load_group(gc);
      wait_group(gc, std::integral_constant<int, 0>{});
      mma_group(gc, sA);
    });
    mx = row_max(sA);
    if constexpr (INIT) {      // a reference exists already (the state's): it moves by the deferred-rescale rule, O and l follow
      if constexpr (VL_SKIP) skip_t0 = all_small(mx);
      if (!__all(mx <= RESCALE_THR)) {
        const float delta = fmaxf(mx, 0.f);
        rescale(sA, delta, __builtin_amdgcn_exp2f(-delta));
      }
    } else {
      rescale(sA, mx, 1.0f);      // first tile: the running max starts at its row maximum
    }
    mx = 0.f;
  }
  __syncthreads();      // K(0) has been read by every wave: iteration 0 overwrites its slot

  int kslot = 0;      // ring slot of the current tile (kt % 3)
  // One key tile.  sc: its scores minus the running max, mx: their row maximum (both from the previous iteration);
  // sn: receives the scores of tile kt+1.  LAST: no next tile; MASKNEXT: tile kt+1 is the last one (masked keys).
  auto iteration = [&](int kt, f16_t (&sc)[2], f16_t (&sn)[2], auto last_c, auto masknext_c, auto vlt_c) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_c)::value;
    constexpr bool MASKNEXT = decltype(masknext_c)::value;
    constexpr bool VLT = decltype(vlt_c)::value;      // with the Vl·P groups (false: this tile's weights are all small, Vh only)
    constexpr int NPV = npv(VLT), NPOS = NG_QK + NPV;
    constexpr int G0 = LAST ? NG_QK : 0;      // the last tile has PV slots only
    // ring slots: K(t) sits in slot t % 3 (ks0), V(t) in slot t % 3 (vs0)
    const int s0 = kslot, s1 = s0 == 2 ? 0 : s0 + 1, s2 = s0 == 0 ? 2 : s0 - 1;      // t%3, (t+1)%3, (t+2)%3
    // staging of K(t+3) -> slot of K(t) (dead since the previous barrier) and V(t+2) -> slot of V(t-1) = (t+2) % 3: one
    // piece every DMA_EVERY slots, between the MFMAs (a piece holds the issuing wave for ~100 cycles)
    constexpr int DMA_EVERY = NPOS / PER_ITER;
    auto stage_piece = [&](int pc) __attribute__((always_inline)) {
#if ZK_ATT_ABL & 8
      if (pc & 1) return;      // (timing probe: half of the staging, wrong results)
#endif
#if !(ZK_ATT_ABL & 1)
      dma_piece(pc, kt + 3, s0, kt + 2, NVS == 4 ? ((kt + 2) & 3) : s2);
#endif
    };
    if constexpr (!LAST) {
      if (!wave_active || ZK_ATT_DMA_TOP) {
#pragma unroll
        for (int pc = 0; pc < PER_ITER; ++pc) stage_piece(pc);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ka[ks] = kofs[ks] + (unsigned)(s1 * KBUF_B);      // K(t+1)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) va[mb] = vofs[mb] + (unsigned)((NVS == 4 ? (kt & 3) : s0) * VBUF_B);       // V(t)

    if (wave_active) {
      // deferred rescale (rare, wave-uniform): before the next tile's scores are started with the running max
      if (!__all(mx <= RESCALE_THR)) {
        const float delta = fmaxf(mx, 0.f);
        rescale(sc, delta, __builtin_amdgcn_exp2f(-delta));
      }
      float psum = 0.f, mxp = -3.0e38f;
      h8_t pf[2][2], pfl_[2][2];      // (pfl_: the lo halves of the weights, PSPLIT only — otherwise never touched and gone after optimisation)
      // (positions p = G0 .. NPOS-1 of the slot order; group id g = seq_g(p); fragment register set p % (LA + 1))
      static_for<LA>([&](auto ic) __attribute__((always_inline)) {
        constexpr int p = G0 + decltype(ic)::value;
        if constexpr (p < NPOS) load_group_at(std::integral_constant<int, seq_g(p, VLT)>{}, std::integral_constant<int, p % (LA + 1)>{});
      });
      if constexpr (LAST) {      // nothing to overlap the exponentials with
#pragma unroll
        for (int e = 0; e < 32; e += 2) psum = exp_pair(sc, pf, pfl_, e, psum);
      }
      __builtin_amdgcn_sched_barrier(0);
      static_for<NPOS - G0>([&](auto ic) __attribute__((always_inline)) {
        constexpr int p = G0 + decltype(ic)::value, g = seq_g(p, VLT);
        using FI = std::integral_constant<int, p % (LA + 1)>;
        if constexpr (!ZK_ATT_DMA_TOP && !LAST && p % DMA_EVERY == DMA_EVERY - 1 && p / DMA_EVERY < PER_ITER) stage_piece(p / DMA_EVERY);
        if constexpr (p + LA < NPOS)
          load_group_at(std::integral_constant<int, seq_g(p + LA, VLT)>{}, std::integral_constant<int, (p + LA) % (LA + 1)>{});
        wait_group_at(std::integral_constant<int, g>{}, FI{}, std::integral_constant<int, seq_reads_after(p, NPOS, VLT)>{});
        if constexpr (g < NG_QK) {
          mma_group_at(std::integral_constant<int, g>{}, FI{}, sn);
#pragma unroll
          for (int e = exp_first(g); e < exp_first(g) + exp_count(g); e += 2) psum = exp_pair(sc, pf, pfl_, e, psum);
        } else {
          constexpr int pvp = p - NG_QK;      // position among this sequence's PV slots (the row-maximum shares go by it)
          constexpr int pv = g - NG_QK, v = VL ? pv / 2 : pv, kb = v / 4, sx = (v / 2) % 2, mb = v % 2;
          constexpr int EPS = 32 / NPV;      // elements of the next tile's row maximum per PV slot
          // (split modes: Vh·P, then Vl·P — the latter only where it can matter: skip_vl is wave-uniform, a scalar branch around
          // one MFMA; its fragment reads stay in the sequence, the counted lgkmcnt waits depend on them)
          if constexpr (VL_SKIP && pv % 2 == 1)
            mfma_vl(oacc[mb], __builtin_bit_cast(h8_t, fr[FI::value].a), pf[kb][sx], skip_vl, std::bool_constant<pv == NG_PV - 1>{});
          else {
            oacc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_t, fr[FI::value].a), pf[kb][sx], oacc[mb], 0, 0, 0);
            if constexpr (PSPLIT && pv % 2 == 0)      // Pl·Vh on the Vh fragment just used
              oacc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_t, fr[FI::value].a), pfl_[kb][sx], oacc[mb], 0, 0, 0);
          }
          if constexpr (!LAST) {
#pragma unroll
            for (int e = EPS * pvp; e < EPS * pvp + EPS; ++e) {
              constexpr int dummy = 0; (void)dummy;
              const int kbn = e >> 4, r = e & 15;
              if constexpr (MASKNEXT) {
                const int key = (NKT - 1) * KT + kbn * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (key >= (GEN ? x.n_keys : S_)) sn[kbn][r] = -1e30f;
              }
              mxp = fmaxf(mxp, sn[kbn][r]);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      l_run += psum;
      if constexpr (!LAST) mx = fmaxf(mxp, __shfl_xor(mxp, 32, 64));
    }   // wave_active

    if constexpr (!LAST) {
      // everything but this iteration's own pieces has landed (K(t+2), V(t+1): what iteration t+1 reads)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ZK_ATT_ABL & 8) ? PER_ITER / 2 : PER_ITER) : "memory");
#if !(ZK_ATT_ABL & 2)
      __builtin_amdgcn_s_barrier();      // (no __syncthreads: its fence would wait for this iteration's pieces as well)
#endif
      kslot = s1;
    }
  };

#if ZK_ATT_STAGGER
  // ---- staggered waves (MI355X guide, "two waves per SIMD", item 9) ----
  // Waves 4-7 — the SIMD partners of waves 0-3 — run the SAME slot sequence half an iteration late: their barrier falls
  // between the score half and the PV half, so the interval between two barriers is  PV(t-1) ‖ max(s_t),  QK(t+1) ‖ exp(s_t)
  // for them while their partners run  QK(t+1) ‖ exp(s_t),  PV(t) ‖ max(s_t+1):  one wave's MFMA-only PV slots beside the
  // other's exp-heavy score slots instead of both waves in the same phase.  Costs: P(t-1) lives across the barrier (it is
  // live anyway), a fourth V ring slot (V(t-1) is read while V(t+2) lands), one more instantiation of the slot sequence.
  h8_t pfr[2][2];      // the late waves' P tile (written in the score half, read in the next interval's PV half)
  h8_t pfrl_[2][2];    // ... and its lo half (PSPLIT only)
  // position p of the late waves' slot order -> group
  // (vlt = false: the Vh groups only, see "the Vl·P pass only where it can matter")
  auto rot_g = [npv](int p, bool vlt) constexpr { return p < npv(vlt) ? NG_QK + ((VL && !vlt) ? 2 * p : p) : p - npv(vlt); };
  auto rot_reads_after = [nreads, rot_g](int p, int n_pos, bool vlt) constexpr {
    int n = 0;
    for (int j = p + 1; j <= p + LA && j < n_pos; ++j) n += nreads(rot_g(j, vlt));
    return n;
  };
  int skip_late = 0;      // late waves: the Vl·P MFMAs of the tile whose PV half comes next are skipped (wave-uniform)
  // the 32-lane exchange of the row maximum as a VALU swap (ds_bpermute would count in lgkmcnt, in the middle of the slots)
  auto xhalf_max = [](float v) __attribute__((always_inline)) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  };
  // One interval of a late wave.  FIRST: interval 0 (score half only: QK(1) ‖ exp(s_0)); FINAL: the part behind the last
  // barrier (PV(NKT-2) ‖ max(s_NKT-1) with the padding keys masked, exp(s_NKT-1), PV(NKT-1)); otherwise interval kt =
  // PV(kt-1) ‖ max(sc = s_kt), [rescale], QK(kt+1) -> sn ‖ exp(sc) -> pfr.
  auto interval_late = [&](int kt, f16_t (&sc)[2], f16_t (&sn)[2], auto first_c, auto final_c, auto vlt_c) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_c)::value, FINAL = decltype(final_c)::value;
    constexpr bool VLT = decltype(vlt_c)::value;          // the PV half of this interval (tile kt-1) with its Vl·P groups
    constexpr int NPV = npv(VLT), NPOS = NG_QK + NPV;
    constexpr int P0 = FIRST ? NPV : 0;                   // first position executed
    constexpr int P1 = FINAL ? NPV : NPOS;                // one past the last
    const int s0 = kslot, s1 = s0 == 2 ? 0 : s0 + 1;      // kt%3, (kt+1)%3
    auto stage_piece = [&](int pc) __attribute__((always_inline)) {
#if !(ZK_ATT_ABL & 1)
      dma_piece(pc, kt + 3, s0, kt + 2, (kt + 2) & 3);
#endif
    };
    if constexpr (!FINAL) {
      if (!wave_active) {
#pragma unroll
        for (int pc = 0; pc < PER_ITER; ++pc) stage_piece(pc);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ka[ks] = kofs[ks] + (unsigned)(s1 * KBUF_B);                // K(kt+1)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) va[mb] = vofs[mb] + (unsigned)(((kt - 1) & 3) * VBUF_B);    // V(kt-1)
    if (wave_active) {
      float psum = 0.f, mxp = -3.0e38f;
      static_for<LA>([&](auto ic) __attribute__((always_inline)) {
        constexpr int p = P0 + decltype(ic)::value;
        if constexpr (p < P1) load_group_at(std::integral_constant<int, rot_g(p, VLT)>{}, std::integral_constant<int, p % (LA + 1)>{});
      });
      __builtin_amdgcn_sched_barrier(0);
      static_for<P1 - P0>([&](auto ic) __attribute__((always_inline)) {
        constexpr int p = P0 + decltype(ic)::value, g = rot_g(p, VLT);
        using FI = std::integral_constant<int, p % (LA + 1)>;
        if constexpr (!FINAL) {
          // the interval's staging pieces, spread over its positions (FIRST has only the score half to put them in)
          constexpr int q = p - P0, every = (P1 - P0) / PER_ITER;
          if constexpr (q % every == every - 1 && q / every < PER_ITER) { if (wave_active) stage_piece(q / every); }
        }
        if constexpr (p + LA < P1)
          load_group_at(std::integral_constant<int, rot_g(p + LA, VLT)>{}, std::integral_constant<int, (p + LA) % (LA + 1)>{});
        wait_group_at(std::integral_constant<int, g>{}, FI{}, std::integral_constant<int, rot_reads_after(p, P1, VLT)>{});
        if constexpr (g >= NG_QK) {
          constexpr int pv = g - NG_QK, v = VL ? pv / 2 : pv, kb = v / 4, sx = (v / 2) % 2, mb = v % 2;
          constexpr int EPS = 32 / NPV;
          if constexpr (VL_SKIP && pv % 2 == 1)
            mfma_vl(oacc[mb], __builtin_bit_cast(h8_t, fr[FI::value].a), pfr[kb][sx], skip_vl, std::bool_constant<pv == NG_PV - 1>{});
          else {
            oacc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_t, fr[FI::value].a), pfr[kb][sx], oacc[mb], 0, 0, 0);
            if constexpr (PSPLIT && pv % 2 == 0)
              oacc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_t, fr[FI::value].a), pfrl_[kb][sx], oacc[mb], 0, 0, 0);
          }
#pragma unroll
          for (int e = EPS * p; e < EPS * p + EPS; ++e) {
            const int kbn = e >> 4, r = e & 15;
            if constexpr (FINAL) {
              const int key = (NKT - 1) * KT + kbn * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
              if (key >= (GEN ? x.n_keys : S_)) sc[kbn][r] = -1e30f;
            }
            mxp = fmaxf(mxp, sc[kbn][r]);
          }
          if constexpr (p == NPV - 1) {      // behind the last PV slot: the running maximum moves before the scores start
            __builtin_amdgcn_sched_barrier(0);
            const float mx = xhalf_max(mxp);
            // does tile kt need its Vl·P groups (next interval's PV half)?  (2^mx against l_run: both relative to m_run)
            if constexpr (VL_SKIP) skip_late = all_small(mx);
            if (!__all(mx <= RESCALE_THR)) {
              const float delta = fmaxf(mx, 0.f);
              rescale(sc, delta, __builtin_amdgcn_exp2f(-delta));
            }
          }
        } else {
          mma_group_at(std::integral_constant<int, g>{}, FI{}, sn);
#pragma unroll
          for (int e = exp_first(g); e < exp_first(g) + exp_count(g); e += 2) psum = exp_pair(sc, pfr, pfrl_, e, psum);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (FINAL) {      // exp(s_NKT-1) with nothing beside it, then PV(NKT-1)
#pragma unroll
        for (int e = 0; e < 32; e += 2) psum = exp_pair(sc, pfr, pfrl_, e, psum);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) va[mb] = vofs[mb] + (unsigned)((kt & 3) * VBUF_B);       // V(kt)
        __builtin_amdgcn_sched_barrier(0);
        auto final_pv = [&](auto v2_c) __attribute__((always_inline)) {      // PV(NKT-1), with or without its Vl·P groups
          constexpr bool V2 = decltype(v2_c)::value;
          constexpr int NP2 = npv(V2);
          static_for<LA>([&](auto ic) __attribute__((always_inline)) {
            constexpr int p = decltype(ic)::value;
            load_group_at(std::integral_constant<int, rot_g(p, V2)>{}, std::integral_constant<int, p % (LA + 1)>{});
          });
          static_for<NP2>([&](auto ic) __attribute__((always_inline)) {
            constexpr int p = decltype(ic)::value, g = rot_g(p, V2);
            using FI = std::integral_constant<int, p % (LA + 1)>;
            if constexpr (p + LA < NP2)
              load_group_at(std::integral_constant<int, rot_g(p + LA, V2)>{}, std::integral_constant<int, (p + LA) % (LA + 1)>{});
            wait_group_at(std::integral_constant<int, g>{}, FI{}, std::integral_constant<int, rot_reads_after(p, NP2, V2)>{});
            constexpr int pv = g - NG_QK, v = VL ? pv / 2 : pv, kb = v / 4, sx = (v / 2) % 2, mb = v % 2;
            if constexpr (VL_SKIP && pv % 2 == 1)
              mfma_vl(oacc[mb], __builtin_bit_cast(h8_t, fr[FI::value].a), pfr[kb][sx], skip_vl, std::bool_constant<pv == NG_PV - 1>{});
            else {
              oacc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_t, fr[FI::value].a), pfr[kb][sx], oacc[mb], 0, 0, 0);
              if constexpr (PSPLIT && pv % 2 == 0)
                oacc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_t, fr[FI::value].a), pfrl_[kb][sx], oacc[mb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
          });
        };
        skip_vl = skip_late;
        final_pv(std::true_type{});
      }
      l_run += psum;
    }
    if constexpr (!FINAL) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_ITER) : "memory");
#if !(ZK_ATT_ABL & 2)
      __builtin_amdgcn_s_barrier();
#endif
      kslot = s1;
    }
  };
#endif
  static_assert((NKT - 1) % 2 == 0, "the tile loop runs in pairs (the two score tiles swap roles)");
  using F = std::false_type;
  using T = std::true_type;
#if ZK_ATT_STAGGER
#ifdef ZK_ATT_STG_PRIO      // probe: static priority for the late (1) or the early (2) half
  if ((ZK_ATT_STG_PRIO == 1) == (wave >= NW / 2)) __builtin_amdgcn_s_setprio(1);
#endif
  if (wave >= NW / 2) {      // (wave-uniform; both branches pass the same NKT-1 barriers)
    // (the PV half of interval kt belongs to tile kt-1: skip_late was decided for it one interval earlier)
    auto late = [&](int kt, f16_t (&sc)[2], f16_t (&sn)[2], auto first_c, auto final_c) __attribute__((always_inline)) {
      skip_vl = skip_late;      // (decided for tile kt-1 one interval earlier; interval_late sets skip_late for tile kt)
      interval_late(kt, sc, sn, first_c, final_c, T{});
    };
    skip_late = skip_t0;
    interval_late(0, sA, sB, T{}, F{}, T{});      // (no PV half)
    for (int kt = 1; kt < NKT - 2; kt += 2) {
      late(kt, sB, sA, F{}, F{});
      late(kt + 1, sA, sB, F{}, F{});
    }
    static_assert((NKT - 3) % 2 == 0, "the late waves' loop ends on interval NKT-3 with the roles (sA, sB)");
    late(NKT - 2, sB, sA, F{}, F{});
    late(NKT - 1, sA, sB, F{}, T{});
  } else
#endif
  {
  // (mx = row maximum of tile kt against the running reference, l_run = the sum over the tiles before it: both known here)
  auto early = [&](int kt, f16_t (&sc)[2], f16_t (&sn)[2], auto last_c, auto masknext_c) __attribute__((always_inline)) {
    if constexpr (VL_SKIP) { if (wave_active) skip_vl = kt == 0 ? skip_t0 : all_small(mx); }
    iteration(kt, sc, sn, last_c, masknext_c, T{});
  };
  for (int kt = 0; kt < NKT - 3; kt += 2) {
    early(kt, sA, sB, F{}, F{});
    early(kt + 1, sB, sA, F{}, F{});
  }
  early(NKT - 3, sA, sB, F{}, F{});
  early(NKT - 2, sB, sA, F{}, T{});
  early(NKT - 1, sA, sB, T{}, F{});
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the surplus pieces of the last iterations)
  // ---- finalize: O / l.  A lane holds 4 consecutive d of ITS query row per register group, i.e. stored directly a wave
  // instruction would write 16-byte fragments of 32 different rows (measured: 10 % of the kernel).  The tile goes
  // through LDS instead (a strip of its own behind the rings) and leaves
  // as whole 128-byte rows, 8 lanes per row.
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  constexpr int O_STR = 144;                       // bytes per staged row (128 + pad: conflict-free 8-byte writes)
  // (the staging strip lies BEHIND the rings: no barrier in front of it, so the early half stores while the late half of
  // a staggered workgroup is still in its last PV slots; aliased into the K ring it cost a barrier and half an iteration
  // of idle early waves per workgroup)
  char* stg = smem + 3 * KBUF_B + NVS * VBUF_B + wave * (32 * O_STR);          // this wave's 32 rows
  // (every per-lane value of the output path derives from a lane index read HERE: computed from the kernel's `lane` they
  // are invariants of the item loop, hipcc hoists them in front of it and spills them across the whole tile loop)
  int ln;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
  const int rd_row = ln >> 3, rd_ch = ln & 7, e_half = ln >> 5, e_row = ln & 31;
  const bool act = wave_active && !DUMP && (!(ZK_ATT_ABL & 16) || lo_fmt == 12345);      // (16: a never-true runtime test keeps the work alive)
  const int row_base = qt * QT + wave * 32;
  if constexpr (DUMP) {      // the running state instead of the normalised output (layer-0 table: constant queries over constant keys)
    const int row = row_base + e_row;
    if (wave_active && row < n_q) {
      float* st = x.state + ((size_t)head * x.state_rows + row) * ST_LD;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          *(f4_t*)(st + 32 * mb + 8 * rg + 4 * e_half) = f4_t{oacc[mb][4 * rg], oacc[mb][4 * rg + 1], oacc[mb][4 * rg + 2], oacc[mb][4 * rg + 3]};
      if (e_half == 0) { st[64] = m_run; st[65] = l_tot; }
    }
  }
  const size_t orow0 = (tok0 + (size_t)row_base) * ZK_HIDDEN + head * ZK_HEAD_DIM;
  const size_t tok0_c = tok0;      // (the item whose rows are about to be stored: set_item below moves on to the next one)
  const int head_c = head;
  bool more = false;
  if constexpr (PERS) {
    // next item: its q loads and K/V prologue pieces go out in front of this item's stores.  The barrier: every wave is past
    // its last fragment read, the rings may be overwritten.
    it_idx += x_per;
    more = it_idx < x_count;
    __builtin_amdgcn_s_barrier();
    if (more) {
      set_item(item_wg(it_idx));
      if constexpr (PERS == 1) q_issue();
      kv_prologue();
    }
  }
  auto flush = [&](half_t* plane) __attribute__((always_inline)) {      // staged rows -> global, 4 x (8 rows x 128 B)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int r = rd_row + 8 * t;
      const h8_t v = *(const h8_t*)(stg + r * O_STR + rd_ch * 16);
#ifdef ZK_ATT_NT
      if (row_base + r < S_) __builtin_nontemporal_store(v, (h8_t*)(plane + orow0 + (size_t)r * ZK_HIDDEN + rd_ch * 8));
#else
      // (o_tiled: the output planes are the O projection's X operand in k-slice-major tiles, zk_planes::tiled — the head's 64
      // columns are one chunk, eight consecutive rows of it 1 KiB contiguous)
      size_t oo;
      bool valid;
      if constexpr (GEN) {      // output row -> token of the window (att_out_token)
        const int row = row_base + r;
        valid = row >= x.q_lo && row < n_q;
        const size_t trow = tok0_c + (size_t)att_out_token(valid ? row : x.q_lo, x.out_map, x.q_lo, x.tr);
        oo = o_tiled ? zk_tiled_off((int)trow, head_c * ZK_HEAD_DIM + rd_ch * 8, ZK_HIDDEN)
                     : trow * ZK_HIDDEN + head_c * ZK_HEAD_DIM + rd_ch * 8;
      } else {
        valid = row_base + r < S_;
        oo = o_tiled ? zk_tiled_off((int)(tok0_c + row_base + r), head_c * ZK_HEAD_DIM + rd_ch * 8, ZK_HIDDEN)
                     : orow0 + (size_t)r * ZK_HIDDEN + rd_ch * 8;
      }
      if (valid) *(h8_t*)(plane + oo) = v;
#endif
    }
  };
  if (act) {
    h4_t lo4[2][4];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int d = 32 * mb + 8 * rg + 4 * e_half;
        h4_t hi;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = oacc[mb][4 * rg + j] * inv;
          zk_pin(v[j]);
          hi[j] = (half_t)v[j];
        }
        *(h4_t*)(stg + e_row * O_STR + d * 2) = hi;
        if (o_lo) lo4[mb][rg] = zk_lo4(v, hi, lo_fmt);
      }
    flush(o_hi);
    if (o_lo) {
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) *(h4_t*)(stg + e_row * O_STR + (32 * mb + 8 * rg + 4 * e_half) * 2) = lo4[mb][rg];
      flush(o_lo);
    }
  }
  if constexpr (PERS) {
    if (!more) break;
    if constexpr (PERS == 2) q_issue();      // (q loads behind the stores: their 32 registers do not overlap the output path's)
  } else {
    break;
  }
  }      // item loop
}

}  // namespace

void zk_launch_attention(zk_planes qkv, zk_planes out, int n_windows, int nsplit, int q_tiles, hipStream_t s, int rev) {
  if (n_windows <= 0) return;
  // q_tiles counts 128-row query blocks (< 10: only the first q_tiles*128 query rows, the last layer's pruning); the
  // kernel's workgroup covers QT = 32·NW rows, waves beyond the row limit only help staging
  // (q_tiles < 0: a limit in ROWS, -q_tiles of them — the pruned last layer asks for one wave's 32)
  if (q_tiles == 0 || q_tiles > 10) q_tiles = 10;
  const int row_limit = q_tiles < 0 ? -q_tiles : q_tiles * 128;
  const int wg_tiles = (row_limit < S_ ? row_limit + QT - 1 : S_ + QT - 1) / QT;
  const int grid = wg_tiles * ZK_HEADS * n_windows;
  auto go = [&](auto kernel, int lds) {
    static bool attr[4] = {false, false, false, false};      // (the 3-deep rings of the split kernels exceed the 64 KiB default)
    if (!attr[nsplit & 3]) {
      (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      attr[nsplit & 3] = true;
    }
#if ZK_ATT_PERSIST
    const int pgrid = grid < 256 ? ((grid + 7) / 8) * 8 : 256;      // one workgroup per CU (LDS), a multiple of the 8 XCDs
#else
    const int pgrid = grid;
#endif
    hipLaunchKernelGGL(kernel, dim3(pgrid), dim3(64 * NW), lds, s, qkv.hi, qkv.lo, out.hi, out.lo, n_windows, wg_tiles, out.lo_fmt, row_limit, rev, out.tiled,
                       att_ext{});
  };
#ifdef ZK_ATT_NO_VL
  constexpr int NIMG_SPLIT = 3;
#else
  constexpr int NIMG_SPLIT = 4;
#endif
  // (K ring of 3 slots x 2 images in the split modes, V ring of NVS slots x (NIMG_SPLIT - 2) images, output staging strip)
  constexpr int STG_B = NW * 32 * 144;
  if (nsplit == 2) go(attention_kernel<2, NKT_FULL, 0>, (3 * 2 + NVS * (NIMG_SPLIT - 2)) * TILE_B + STG_B);
  else if (nsplit == 3) go(attention_kernel<3, NKT_FULL, 0>, (3 * 2 + NVS * (NIMG_SPLIT - 2)) * TILE_B + STG_B);
  else go(attention_kernel<1, NKT_FULL, 0>, (3 + NVS) * TILE_B + STG_B);
}

// layer-0 constant-row attention: the three generalised launches (zk_common.h)
void zk_launch_attention_l0(int what, zk_planes ctab, float* state, zk_planes tail, zk_planes out, int n_windows, int t_real,
                            int nsplit, hipStream_t s) {
  if (what != ZK_L0_ATT_DUMP && n_windows <= 0) return;
  const int n_real = ZK_FOUT * t_real, n_const = S_ - n_real, rem = n_const - 1088;
  att_ext x{};
  x.a_hi = ctab.hi; x.a_lo = ctab.lo; x.state = state; x.tr = t_real; x.state_rows = n_const; x.b_rows = 128;
  int nq, nw = n_windows;
  if (what == ZK_L0_ATT_DUMP) {           // 17 tiles of the table, keys 0..1023 valid (the 17th tile is masked: odd tile count)
    x.seg = 17; x.a_row0 = 0; x.n_keys = 1024; x.q_from_a = 1; x.a_q_row0 = 0; x.n_q = nq = n_const; x.q_lo = 0; x.out_map = 1; nw = 1;
  } else if (what == ZK_L0_ATT_CONST) {   // table rows 1024..1087, then the window's tail: rem constant + n_real real keys
    x.seg = 1; x.a_row0 = 1024; x.n_keys = 64 + rem + n_real; x.q_from_a = 1; x.a_q_row0 = 0; x.n_q = nq = n_const; x.q_lo = 0; x.out_map = 1;
  } else {                                // all keys in the order [constant | real]; the queries are the tail's real rows
    x.seg = 17; x.a_row0 = 0; x.n_keys = S_; x.q_from_a = 0; x.n_q = nq = rem + n_real; x.q_lo = rem; x.out_map = 2;
  }
  const int wg_tiles = (nq + QT - 1) / QT;
  const int grid = wg_tiles * ZK_HEADS * nw;
#ifdef ZK_ATT_NO_VL
  constexpr int NIMG_SPLIT = 3;
#else
  constexpr int NIMG_SPLIT = 4;
#endif
  constexpr int STG_B = NW * 32 * 144, LDS = (3 * 2 + NVS * (NIMG_SPLIT - 2)) * TILE_B + STG_B;
  // (probe ZK_ATT_PERSIST_L0: the constant-query launch persistent — one workgroup per CU (LDS), a multiple of the 8 XCDs)
  const bool pers = what == ZK_L0_ATT_CONST && ZK_ATT_PERSIST_L0 != 0;
  const int pgrid = pers ? (grid < 256 ? ((grid + 7) / 8) * 8 : 256) : grid;
  auto go = [&](auto kernel) {
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipLaunchKernelGGL(kernel, dim3(pgrid), dim3(64 * NW), LDS, s, tail.hi, tail.lo, out.hi, out.lo, nw, wg_tiles, out.lo_fmt, nq, 0, out.tiled, x);
  };
  {
  if (nsplit == 2) {
    if (what == ZK_L0_ATT_DUMP) go(attention_kernel<2, 17, 3>);
    else if (what == ZK_L0_ATT_CONST) go(attention_kernel<2, 3, 2>);
    else go(attention_kernel<2, NKT_FULL, 1>);
  } else {
    if (what == ZK_L0_ATT_DUMP) go(attention_kernel<3, 17, 3>);
    else if (what == ZK_L0_ATT_CONST) go(attention_kernel<3, 3, 2>);
    else go(attention_kernel<3, NKT_FULL, 1>);
  }
  }
}
