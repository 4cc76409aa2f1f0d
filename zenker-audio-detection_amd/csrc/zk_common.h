// Shared declarations for the zkast HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef float f16_t __attribute__((ext_vector_type(16)));
typedef short s4v_t __attribute__((__vector_size__(4 * sizeof(short))));

// AST geometry (ASTConfig defaults, $TF/models/audio_spectrogram_transformer/configuration_...py:50-64).
// The library validates a loaded config against these and refuses anything else.
#define ZK_HIDDEN 768
#define ZK_HEADS 12
#define ZK_HEAD_DIM 64
#define ZK_INTER 3072
#define ZK_LAYERS 12
#define ZK_PATCH 16
#define ZK_FSTRIDE 10
#define ZK_TSTRIDE 10
#define ZK_NMEL 128
#define ZK_MAXLEN 1024
#define ZK_FOUT 12
#define ZK_TOUT 101
#define ZK_NPATCH 1212
#define ZK_SEQ 1214
#define ZK_PATCH_K 256

// log-mel geometry ($TF/.../feature_extraction_audio_spectrogram_transformer.py:124-141)
#define ZK_FRAME_LEN 400
#define ZK_FRAME_HOP 160
#define ZK_FFT 512
#define ZK_NBINS 257
#define ZK_MEL_BAND_MAX 1024   // LDS slots for the mel bank's band (sum over filters of mel_hi - mel_lo; 504 for the AST bank)

// activation planes: a tensor is stored as one (hi) or two (hi, lo) 16-bit planes.
//   lo_fmt ZK_LO_F16: lo = fp16(x - hi), x ~= hi + lo                      (ZK_F16X3 GEMMs, split QK^T)
//   lo_fmt ZK_LO_C8 : lo = (fp8 e4m3 of (x - hi)·2^11, fp8 e4m3 of x) byte pair   (ZK_F16C8 GEMMs: the two correction
//                     products of the split run on the fp8 matrix pipe as ONE K'=2K product, see gemm_c8.hip)
enum { ZK_LO_F16 = 0, ZK_LO_C8 = 1 };
struct zk_planes {
  half_t* hi;
  half_t* lo;  // nullptr in single-pass mode
  int lo_fmt;
  // ZK_F16C8 activations only, optional: per-row power-of-two exponent s_m.  Both planes of row m then hold x·2^-s_m
  // (largest |x| of the row in (112, 224]: the fp8 bytes use e4m3's whole range whatever the row's magnitude, nothing
  // saturates at 448 and small rows keep their correction) and the consuming GEMM multiplies its accumulator row by
  // 2^s_m in the epilogue — exact, and free in the k-loop.  nullptr: planes are unscaled.
  int32_t* rowexp;
  // ZK_F16C8 GEMM operands only: the planes lie in k-slice-major tiles — [row block of 256][64-column chunk][256 rows][128 B],
  // the eight 16-byte chunks of a row in the LDS image's swizzled order (chunk c of row m at position c ^ ((m >> 1) & 7)) —
  // so that the X half of a GEMM ring step is 32 KiB contiguous in HBM (gemm_c8.hip).  Planes must hold whole row blocks.
  int tiled = 0;
  // rows the planes are allocated for (set by the owner of the buffers; 0 = not stated).  zk_launch_gemm_c8 refuses an
  // x operand whose last 256-row block is not wholly inside the allocation.
  int64_t rows_cap = 0;
};
#ifdef __HIPCC__
// element offset of columns [c, c + 8) ∩ one 16-byte chunk of row m in a tiled plane with K columns
__device__ __forceinline__ size_t zk_tiled_off(int m, int c, int K) {
  return (((size_t)(m >> 8) * (K >> 6) + (c >> 6)) * 256 + (m & 255)) * 64 + (((((c & 63) >> 3) ^ ((m >> 1) & 7))) << 3) + (c & 7);
}
#endif

#define ZK_C8_SHIFT 11   // (x - hi) is scaled by 2^11 before the fp8 rounding: |x - hi| <= 2^-11 |x|

#ifdef __HIPCC__
// exponent s of a row whose largest magnitude is amax: amax·2^-s lies in (112, 224] (0 for an all-zero row)
__device__ __forceinline__ int zk_row_exponent(float amax) {
  if (!(amax > 0.f) || !(amax < 3.0e38f)) return 0;
  int e;
  const float f = frexpf(amax, &e);      // amax = f·2^e, f in [0.5, 1)
  int s = (f <= 0.875f) ? e - 8 : e - 7; // 224 = 0.875·2^8
  return s < -40 ? -40 : (s > 40 ? 40 : s);
}
// clamp to e4m3's finite range: v_cvt_pk_fp8_f32 does NOT saturate (probed on gfx950: 448 < |x| <= 464 rounds to 448, beyond
// that the byte is NaN).  v_med3_f32 directly: the fminf(fmaxf()) form costs a second instruction (hipcc quiets a possible
// signalling NaN with v_max x,x first).
__device__ __forceinline__ float zk_clamp_fp8(float x) { return __builtin_amdgcn_fmed3f(x, -448.f, 448.f); }
// two consecutive elements (values v0, v1; h01 = their fp16 roundings, packed) -> one dword of two (lo8, x8) byte pairs
// (OCP e4m3, RNE).  v - fp16(v) comes from ONE v_fma_mix_f32 that reads the fp16 half in place (the difference is exact in
// fp32 either way), and the first conversion writes into whatever `old` holds — its upper half is overwritten by the second.
__device__ __forceinline__ unsigned zk_c8_pack2(float v0, float v1, h2_t h01) {
  float l0, l1;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(h01), "v"(v0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(h01), "v"(v1));
  int p = __builtin_amdgcn_cvt_pk_fp8_f32(zk_clamp_fp8(l0 * 2048.f), zk_clamp_fp8(v0), __builtin_bit_cast(int, v0), false);
  p = __builtin_amdgcn_cvt_pk_fp8_f32(zk_clamp_fp8(l1 * 2048.f), zk_clamp_fp8(v1), p, true);
  return (unsigned)p;
}
// Pin a computed fp32 value before its (hi, lo) planes are derived from it.  hipcc contracts across statements
// (-ffp-contract=fast): (half_t)(a * b) may become ONE v_fma_mixlo_f16 — a single rounding of the exact product — in one
// use and fp16(fp32(a * b)) in another, and a lo entry computed against a different hi than the one that was stored is off
// by a whole fp16 ulp (seen in round 3: attention's fp16 lo plane, 1.3e-4 -> 4.5e-4 of max|ref| after an unrelated edit
// of this header).  The empty asm makes the value opaque: every later use sees the same rounded fp32 number.
__device__ __forceinline__ void zk_pin(float& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void zk_pin(f4_t& x) { asm volatile("" : "+v"(x)); }
// lo plane entries of 4 consecutive elements in either format
__device__ __forceinline__ h4_t zk_lo4(const float* v, h4_t hi, int fmt) {
  if (fmt == ZK_LO_C8) {
    unsigned d[2] = {zk_c8_pack2(v[0], v[1], h2_t{hi[0], hi[1]}), zk_c8_pack2(v[2], v[3], h2_t{hi[2], hi[3]})};
    return __builtin_bit_cast(h4_t, d);
  }
  // fp16 lo = fp16(v - hi): v_fma_mixlo_f16 / _mixhi_f16 read the fp16 half in place and round the exact difference once —
  // one instruction per element instead of convert, subtract, convert
  unsigned d[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const h2_t h01 = {hi[2 * p], hi[2 * p + 1]};
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d[p]) : "v"(h01), "v"(v[2 * p]));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d[p]) : "v"(h01), "v"(v[2 * p + 1]));
  }
  return __builtin_bit_cast(h4_t, d);
}
#endif

// ---- GEMM epilogues -------------------------------------------------------------------------------
enum { ZK_EPI_STORE = 0, ZK_EPI_GELU = 1, ZK_EPI_RESID = 2, ZK_EPI_PATCH = 3 };

struct zk_gemm_args {
  // x planes: zk_launch_gemm_c8 stages the last row block WHOLE, so both planes must be readable up to row
  // ceil(M/256)*256 (the workspace and the test hooks allocate one spare tile; what those rows hold never reaches a store)
  const half_t* x_hi;  // [M, K] activations (row-major, K contiguous)
  const half_t* x_lo;  // or nullptr
  const half_t* w_hi;  // [N, K] weights (nn.Linear layout)
  const half_t* w_lo;
  const float* bias;   // [N]
  const int32_t* x_rowexp;  // [M] or nullptr: row m of the x planes holds x·2^-x_rowexp[m] (zk_planes::rowexp)
  int M, N, K;
  int ldo = 0;         // STORE / GELU, row-major output planes: elements between two output rows (the launchers set N when 0);
                       // > N writes an N-column band of wider planes (the last layer's k|v-only QKV launch)
  // outputs
  half_t* o_hi;        // [M, N] (STORE / GELU)
  half_t* o_lo;
  float* resid;        // [M, N] fp32, in-place += (RESID) ; PATCH: hidden base
  const float* pos;    // PATCH: position embeddings [1214, 768]
  int lo_n_limit;      // STORE: write the lo plane only for n < lo_n_limit
  int lo_c8_from;      // STORE (ZK_F16C8): columns lo_c8_from <= n < min(lo_c8_to, lo_n_limit) get a c8 lo plane (k of the fused
  int lo_c8_to = 1 << 30;   //   QKV: the fp8-corrected QK^T), the others an fp16 lo plane (q: re-split by attention; v: the Vl·P pass)
  int w_exp;           // ZK_F16C8: the weight's c8 plane holds (fp8(W·2^w_exp), fp8((W-Wh)·2^(w_exp+11)))
  int rev = 0;         // ZK_F16C8: walk the row blocks from the last to the first (same results, other order)
  int64_t x_rows = 0;  // ZK_F16C8: rows the x planes are ALLOCATED for; must cover ceil(M/256)*256 (checked by the launcher)
  int x_tiled = 0;     // ZK_F16C8: the x planes are k-slice-major tiles (zk_planes::tiled)
  int o_tiled = 0;     // ZK_F16C8, GELU epilogue: write the output planes that way (the operand of the next GEMM)
  // PATCH epilogue: 0 = the A rows are every patch of every window ([b][f][t], 1212 per window); tr > 0: only the time
  // patches t < tr ([b][f][t < tr], 12·tr per window) — the layer-0 constant-row reuse (zkast.hip) computes the others once
  int patch_tr = 0;
};

// launchers (each file owns its kernels)
void zk_launch_gemm(const zk_gemm_args& a, int epi, int nsplit, hipStream_t s);
// rev (LayerNorm, attention, zk_gemm_args::rev): process the rows from the last to the first — same results; lets a
// kernel start on what its producer wrote last (zkast.hip: walk alternation)
// gather_tr > 0 (layer-0 constant-row reuse): `rows` counts COMPACT rows [b][f][t < gather_tr]; compact row j is read from
// row b·1214 + 2 + f·101 + t of x and written to row j of the planes
void zk_launch_layernorm(const float* x, int64_t row_stride, const float* gamma, const float* beta, int rows,
                         zk_planes out, float eps, hipStream_t s, int rev = 0, int gather_tr = 0);
void zk_launch_attention(zk_planes qkv, zk_planes out, int n_windows, int nsplit, int q_tiles, hipStream_t s, int rev = 0);
void zk_launch_gather_tok01(zk_planes att, const float* hidden, int n_windows, zk_planes att_out, float* hidden_out,
                            hipStream_t s);
// t_real > 0: only the time patches t < t_real ([b][f][t < t_real] rows, 12·t_real per window)
void zk_launch_im2col_compact(const float* feats, int n_frames, const int32_t* win_idx, int n_windows, float mean,
                              float std2, zk_planes out, hipStream_t s, int t_real = 0);
// layer-0 constant-row reuse (zkast.hip): the rows of a window whose layer-0 input does not depend on the window — cls,
// distillation and every patch token with t >= t_real (it sees only the extractor's padding) — are copied from a table
// computed once per model: the fp32 residual rows, and the layer-0 q|k|v planes together with the freshly computed rows
// of the real tokens (compact [b][f][t < t_real] planes)
// last-layer pruning, query side: LayerNorm rows 0 .. ZK_QROWS-1 of every window -> compact planes [ZK_QROWS·n_windows, 768]
// (row exponents included), and the compact q rows back into columns 0..767 of those rows of the windows' q|k|v planes
#define ZK_QROWS 32      // = the query rows of one attention wave (attention.hip: q_row = tile·256 + wave·32 + lane % 32)
void zk_launch_gather_xq(zk_planes x, int n_windows, zk_planes out, hipStream_t s);
void zk_launch_scatter_q(zk_planes q, int n_windows, zk_planes qkv, hipStream_t s);
void zk_launch_l0_fill_hidden(float* hidden, const float* table, int n_windows, int t_real, hipStream_t s);
void zk_launch_l0_assemble_qkv(zk_planes real_rows, zk_planes table, zk_planes out, int n_windows, int t_real, hipStream_t s);
// ---- layer-0 constant-row ATTENTION (attention.hip / embed.hip; orchestrated by zkast.hip) ----
// With t_real real time patches per frequency row a window has n_real = 12·t_real real tokens and n_const = 1214 - n_real
// constant ones.  The scores of constant queries against constant keys do not depend on the window, so their running softmax
// state (O unnormalised, m, l) over the first 1024 constant keys is tabulated once per model and every window only adds
//   * for its constant queries: constant keys 1024 .. n_const-1 and its real keys            (ZK_L0_ATT_CONST, 3 key tiles),
//   * for its real queries: all keys, in the order [constant | real]                       (ZK_L0_ATT_REAL, 19 key tiles),
// 0.27 M instead of 1.47 M query-key pairs per window and head.  Key tiles 0..16 are rows of the model's constant-order table
// (shared by all windows), tiles 17..18 the window's 128-row TAIL planes (n_const - 1088 constant rows, the real rows, zero
// padding).  Needs 1088 <= n_const < 1152 and n_const - 1088 + n_real <= 126, i.e. 6 <= t_real <= 10 (zk_l0_att_supported).
#define ZK_L0_CTAB_ROWS 1152      // rows of the constant-order table planes (18 whole key tiles; rows >= n_const are zero)
#define ZK_L0_STATE_LD 68         // floats per state row: O[64], m, l, 2 pad
enum { ZK_L0_ATT_DUMP = 0, ZK_L0_ATT_CONST = 1, ZK_L0_ATT_REAL = 2 };
inline bool zk_l0_att_supported(int t_real) { return t_real >= 6 && t_real <= 10; }
void zk_launch_l0_const_order(zk_planes table, zk_planes out, int t_real, int rows_pad, hipStream_t s);
void zk_launch_l0_tail(zk_planes real_rows, zk_planes ctab, zk_planes out, int n_windows, int t_real, hipStream_t s);
// what = ZK_L0_ATT_DUMP: state of the constant queries over constant keys 0..1023 -> `state` (ctab only; n_windows ignored);
// ZK_L0_ATT_CONST / _REAL: the two per-forward launches, output rows scattered to their tokens in `out` ([M, 768] planes)
void zk_launch_attention_l0(int what, zk_planes ctab, float* state, zk_planes tail, zk_planes out, int n_windows, int t_real,
                            int nsplit, hipStream_t s);
void zk_launch_im2col_full(const float* input_values, int n_windows, zk_planes out, hipStream_t s);
void zk_launch_cls_rows(float* hidden, const float* cls, const float* dist, const float* pos, int n_windows,
                        hipStream_t s);
void zk_launch_head(const float* hidden, int rows_per_window, int n_windows, const float* lnf_g, const float* lnf_b, const float* lnh_g,
                    const float* lnh_b, const float* w, const float* b, int num_labels, float eps, float* logits,
                    hipStream_t s);
void zk_launch_logmel(const float* audio, int64_t n_samples, int64_t first_start, int64_t hop, int32_t win,
                      int n_windows, int n_frames, const double* hann, const double* twiddle, const double* mel,
                      const int32_t* mel_lo, const int32_t* mel_hi, float* out, hipStream_t s);
void zk_launch_expand_features(const float* feats, int n_frames, int n_windows, float mean, float std2,
                               int do_normalize, float* out, hipStream_t s);
void zk_launch_gate(const float* logits, int n, float thr1, float fwd_min_prob, float* probs, int32_t* idx,
                    int32_t* count, hipStream_t s);
void zk_launch_softmax2(const float* logits, int n, int num_labels, float* probs, hipStream_t s);
void zk_launch_resample(const float* in, int64_t n_in, int orig, int neu, int width, const float* kernels,
                        int klen, float* out, int64_t n_out, hipStream_t s);
void zk_launch_wav_decode(const unsigned char* raw, int64_t n_frames, int fmt, int bits, int channels, float* out,
                          hipStream_t s);
void zk_launch_split_f32(const float* src, int64_t n, float scale, half_t* hi, half_t* lo, hipStream_t s);
// weights: c8 plane = (fp8(w·2^e), fp8((w - fp16(w))·2^(e+11))) byte pairs; activations (is_weight = 0): (fp8((x-xh)·2^11), fp8(x))
void zk_launch_split_c8(const float* src, int64_t n, int w_exp, int is_weight, half_t* c8, hipStream_t s);
// activations [rows, K] fp32 -> row-scaled fp16 + c8 planes and the row exponents (zk_planes::rowexp); K % 4 == 0
// columns [c0, c0 + ncols) of a [rows, ld] fp32 matrix -> activation c8 entries at the same positions of `lo` (ncols % 4 == 0)
void zk_launch_split_c8_cols(const float* src, int rows, int ld, int c0, int ncols, half_t* lo, hipStream_t s);
void zk_launch_split_rows_c8(const float* src, int rows, int K, half_t* hi, half_t* c8, int32_t* rowexp, hipStream_t s);
// returns 0, or -1 (nothing launched) when the shape / padding contract is violated: N % 256, K % 64, M < 1, or x planes
// that do not cover the last row block whole (zk_gemm_args::x_rows)
int zk_launch_gemm_c8(const zk_gemm_args& a, int epi, hipStream_t s);
