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

// activation planes: a tensor is stored as one (hi) or two (hi, lo) fp16 planes; x ~= hi + lo
struct zk_planes {
  half_t* hi;
  half_t* lo;  // nullptr in single-pass mode
};

// ---- GEMM epilogues -------------------------------------------------------------------------------
enum { ZK_EPI_STORE = 0, ZK_EPI_GELU = 1, ZK_EPI_RESID = 2, ZK_EPI_PATCH = 3 };

struct zk_gemm_args {
  const half_t* x_hi;  // [M, K] activations (row-major, K contiguous)
  const half_t* x_lo;  // or nullptr
  const half_t* w_hi;  // [N, K] weights (nn.Linear layout)
  const half_t* w_lo;
  const float* bias;   // [N]
  int M, N, K;
  // outputs
  half_t* o_hi;        // [M, N] (STORE / GELU)
  half_t* o_lo;
  float* resid;        // [M, N] fp32, in-place += (RESID) ; PATCH: hidden base
  const float* pos;    // PATCH: position embeddings [1214, 768]
  int lo_n_limit;      // STORE: write the lo plane only for n < lo_n_limit
  long long* stamps;   // diagnostic only (ZK_GEMM_STAMPS): [grid][16] s_memtime stamps, nullptr in production
  int ablate;          // diagnostic only (ZK_GEMM_ABLATE): timing-only knobs, 0 in production
};

// launchers (each file owns its kernels)
void zk_launch_gemm(const zk_gemm_args& a, int epi, int nsplit, hipStream_t s);
void zk_launch_layernorm(const float* x, int64_t row_stride, const float* gamma, const float* beta, int rows,
                         zk_planes out, float eps, hipStream_t s);
void zk_launch_attention(zk_planes qkv, zk_planes out, int n_windows, int nsplit, int q_tiles, hipStream_t s);
void zk_launch_gather_tok01(zk_planes att, const float* hidden, int n_windows, zk_planes att_out, float* hidden_out,
                            hipStream_t s);
void zk_launch_im2col_compact(const float* feats, int n_frames, const int32_t* win_idx, int n_windows, float mean,
                              float std2, zk_planes out, hipStream_t s);
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
void zk_launch_split_f32(const float* src, int64_t n, float scale, half_t* hi, half_t* lo, hipStream_t s);
