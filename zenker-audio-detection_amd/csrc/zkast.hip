// libzkast.so — context, model store, forward orchestration and the C ABI declared in include/zkast.h.
// Host code only (the kernels live in gemm/attention/layernorm/embed/head/logmel/misc .hip).
#include "../../include/zkast.h"
#include "zk_common.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

namespace {

thread_local std::string g_create_err;

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    // round up so that slowly growing requests do not re-allocate every call
    size_t want = bytes + (bytes >> 3) + 4096;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { e = hipMalloc(&p, bytes); want = bytes; }
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <class T> T* as() const { return (T*)p; }
};

// Device allocations of one test-hook call: freed when the hook returns, on EVERY path (the HIPCHK / fail() early
// returns included).  The stream is drained first so that no queued kernel still reads a buffer being freed.
struct DevArena {
  hipStream_t stream;
  std::vector<void*> ptrs;
  explicit DevArena(hipStream_t s) : stream(s) {}
  DevArena(const DevArena&) = delete;
  DevArena& operator=(const DevArena&) = delete;
  template <class T> hipError_t alloc(T** out, size_t bytes) {
    void* p = nullptr;
    const hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    *out = (T*)p;
    if (e == hipSuccess) ptrs.push_back(p);
    return e;
  }
  ~DevArena() {
    if (!ptrs.empty()) (void)hipStreamSynchronize(stream);
    for (void* p : ptrs) (void)hipFree(p);
  }
};

struct PlaneBuf {
  DevBuf hi, lo, rowexp;      // rowexp: allocated only for the LayerNorm outputs (zk_planes::rowexp)
  zk_planes get(bool split, int lo_fmt = ZK_LO_F16) const {
    zk_planes pl{hi.as<half_t>(), split ? lo.as<half_t>() : nullptr, lo_fmt,
                 (split && lo_fmt == ZK_LO_C8) ? rowexp.as<int32_t>() : nullptr};
    pl.rows_cap = (int64_t)((split && lo.cap < hi.cap ? lo.cap : hi.cap) / 2);      // in ELEMENTS here; run_gemm divides by K
    return pl;
  }
};

// a weight matrix in all the forms the GEMM modes read: fp16 hi (ZK_F16), + fp16 lo (ZK_F16X3), + c8 byte pairs with
// their power-of-two exponent (ZK_F16C8); 6 B per parameter, 0.5 GB per stage
struct WMat {
  half_t *hi = nullptr, *lo = nullptr, *c8 = nullptr;
  int exp = 0;
};

struct LayerW {
  float *ln1_g, *ln1_b, *ln2_g, *ln2_b, *bqkv, *bo, *b1, *b2;
  WMat wqkv, wo, w1, w2;
};

// ZK_F16MIX: one nibble per encoder layer (layer l = bits 4l .. 4l+3: QKV GEMM, QK^T, O projection, MLP); a set bit runs that
// kernel group in the 3-pass ZK_F16X3 arithmetic, a clear one in ZK_F16C8 (chosen on the input-sensitive
// weight set so that a configs[3]-sized recording keeps >= 20 % of the 1e-3 logit tolerance)
#ifndef ZK_MIX_X3_MASK
#define ZK_MIX_X3_MASK 0x3ull      // layer 0: QKV, QK^T (zkast/lib.py: MIX_X3_GROUPS)
#endif
struct LayerMode { int qkv, att, o, mlp; };
// attention reads k's lo plane as c8 byte pairs only when the ZK_F16C8 QKV epilogue wrote them, and any lo plane only if
// the QKV GEMM wrote one
inline bool layer_mode_ok(const LayerMode& m) {
  auto known = [](int v) { return v == ZK_F16 || v == ZK_F16C8 || v == ZK_F16X3; };
  if (!known(m.qkv) || !known(m.att) || !known(m.o) || !known(m.mlp)) return false;
  if (m.att == ZK_F16C8 && m.qkv != ZK_F16C8) return false;
  if (m.att == ZK_F16X3 && m.qkv == ZK_F16) return false;
  return true;
}
struct StageModel {
  bool loaded = false;
  int mode = ZK_F16X3;            // what the caller asked for (ZK_F16MIX: see layer_mode)
  // the mode each kernel group of each encoder layer runs in (ZK_F16 / ZK_F16C8 / ZK_F16X3): the fused QKV GEMM, attention's
  // QK^T, the O projection, the MLP (FC1 + FC2 share a mode: FC1's epilogue writes FC2's operand planes).  Layers only
  // exchange the fp32 residual stream and every producer writes the plane format its consumer reads, so any assignment
  // that passes layer_modes_ok() is a valid forward (zk_model_set_layer_modes)
  LayerMode layer_mode[ZK_LAYERS];
  // mode of everything outside the encoder layers (the patch embedding): ZK_F16MIX counts as ZK_F16C8
  int base_mode() const { return mode == ZK_F16MIX ? ZK_F16C8 : mode; }
  bool any_split() const {
    for (int l = 0; l < n_layers; ++l) { const LayerMode& m = layer_mode[l]; if (m.qkv != ZK_F16 || m.att != ZK_F16 || m.o != ZK_F16 || m.mlp != ZK_F16) return true; }
    return base_mode() != ZK_F16;
  }
  void set_mode(int m) {
    mode = m;
    for (int l = 0; l < ZK_LAYERS; ++l) {
      if (m != ZK_F16MIX) { layer_mode[l] = LayerMode{m, m, m, m}; continue; }
      const int g = (ZK_MIX_X3_MASK >> (4 * l)) & 15;      // bit 0 QKV, 1 QK^T, 2 O, 3 MLP
      layer_mode[l] = LayerMode{g & 1 ? ZK_F16X3 : ZK_F16C8, g & 2 ? ZK_F16X3 : ZK_F16C8, g & 4 ? ZK_F16X3 : ZK_F16C8, g & 8 ? ZK_F16X3 : ZK_F16C8};
    }
  }
  int num_labels = 2;
  int n_layers = ZK_LAYERS;
  float eps = 1e-12f;
  float mean = 0.f, std = 1.f;
  std::vector<void*> allocs;
  float *cls = nullptr, *dist = nullptr, *pos = nullptr, *patch_b = nullptr;
  WMat patch_w;
  LayerW L[ZK_LAYERS];
  float *lnf_g = nullptr, *lnf_b = nullptr, *lnh_g = nullptr, *lnh_b = nullptr, *head_w = nullptr, *head_b = nullptr;
  // layer-0 constant-row table (build_l0_table): the fp32 residual rows and the layer-0 q|k|v planes of ONE window; valid
  // for the rows that do not depend on the window, given (n_frames, compute mode, fx mean / std) — l0_frames < 0: not built
  float* l0_hidden = nullptr;
  half_t *l0_qkv_hi = nullptr, *l0_qkv_lo = nullptr;
  int l0_frames = -1, l0_mode = 0, l0_base = 0;
  // layer-0 constant-row attention (zk_common.h): the table's q|k|v rows in constant-token order and the running softmax state of
  // the constant queries over the first 1024 constant keys; l0_att: valid for the current table
  half_t *l0_ctab_hi = nullptr, *l0_ctab_lo = nullptr;
  float* l0_state = nullptr;
  bool l0_att = false;
  float l0_mean = 0.f, l0_std = 0.f;
  void release() {
    for (void* p : allocs) (void)hipFree(p);
    allocs.clear();
    loaded = false;
    l0_hidden = nullptr; l0_qkv_hi = l0_qkv_lo = nullptr; l0_frames = -1;
    l0_ctab_hi = l0_ctab_lo = nullptr; l0_state = nullptr; l0_att = false;
  }
};

enum ProfClass { P_GEMM_QKV, P_GEMM_O, P_GEMM_FC1, P_GEMM_FC2, P_GEMM_PATCH, P_ATTN, P_LN, P_LOGMEL, P_EMBED, P_HEAD, P_WAVDEC,
                 P_RESAMPLE, P_ALLGATHER,
                 // ZK_F16MIX only: the launches of the layers that run ZK_F16X3 are other kernels than the ZK_F16C8 ones and
                 // are timed apart, so that each class stays ONE kernel instantiation (bench.py's roofline)
                 P_GEMM_QKV_X3, P_GEMM_O_X3, P_GEMM_FC1_X3, P_GEMM_FC2_X3, P_ATTN_X3, P_N };
const char* kProfNames[P_N] = {"gemm_qkv", "gemm_o", "gemm_fc1", "gemm_fc2", "gemm_patch",
                               "attention", "layernorm", "logmel", "embed", "head", "wav_decode", "resample", "allgather",
                               "gemm_qkv_x3", "gemm_o_x3", "gemm_fc1_x3", "gemm_fc2_x3", "attention_x3"};
// class of a layer kernel: inside a ZK_F16MIX model the ZK_F16X3 layers count apart


}  // namespace

struct zk_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  bool async = false;
  int micro_batch = 0;   // 0 = auto
  // Walk alternation (ZK_WALK_ALT=1 in the environment at zk_create): consecutive kernels of a layer process the token rows
  // in opposite directions, so each starts on the rows its producer wrote last (the tail of a 2-15 GB plane that may
  // still be in the 256 MiB Infinity Cache) instead of the rows that left the caches first.  Results are identical.
  bool walk_alt = false;
  int walk_dir = 0;
  std::string err;
  StageModel model[2];

  // feature-extraction tables (fp64) and slot
  double *d_hann = nullptr, *d_tw = nullptr, *d_mel = nullptr;
  int32_t *d_mel_lo = nullptr, *d_mel_hi = nullptr;
  DevBuf feat;  // compact [n_windows, n_frames, 128] fp32
  int feat_windows = 0, feat_frames = 0;

  // audio slot: the recording load_audio left on the device (zk_audio_load)
  DevBuf audio_slot;
  int64_t audio_slot_n = 0;

  // staging + workspace
  DevBuf st_in, st_out, st_idx, audio_dev, s1_logits, s2_logits, gate_idx, gate_cnt, tmp_f32;
  DevBuf hidden;
  PlaneBuf patchA, xn, qkv, att, mid, att_s, xn_s, mid_s;   // *_s: tokens 0/1 only (last-layer pruning)
  DevBuf hidden_s;
  DevBuf xq_rowexp;      // row exponents of the gathered LayerNorm rows of the pruned last layer's q GEMM (ZK_QROWS per window)
  bool prune_last = true;
  bool l0_reuse = true;      // layer-0 constant-row reuse (zk_set_layer0_reuse)
  bool l0_attention = true;  // ... including the constant-row attention state (zk_set_layer0_attention)
  int ws_windows = 0;
  bool ws_split = false;

  // resampler kernel cache
  DevBuf rs_kern;
  int rs_orig = 0, rs_new = 0, rs_width = 0, rs_klen = 0;

  // profiling
  bool prof = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  std::vector<std::pair<int, int>> ev_used;  // (class, pool index)
  double prof_ms[P_N] = {0};
  double prof_flops[P_N] = {0};
  int64_t prof_n[P_N] = {0};

  // multi-GPU: RCCL communicator (comm.hip owns the object) and its host<->device staging
  void* comm = nullptr;
  DevBuf comm_stage[2];

  // debug tap
  int tap_layer = -2;
  DevBuf tap;
  int tap_windows = 0;
};

namespace {

int fail(zk_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_create_err = buf;
  return code;
}

#define HIPCHK(c, expr)                                                                                   \
  do {                                                                                                    \
    hipError_t e__ = (expr);                                                                              \
    if (e__ != hipSuccess) return fail((c), ZK_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                                       __FILE__, __LINE__);                                               \
  } while (0)

bool is_device_ptr(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t a;
  memset(&a, 0, sizeof a);
  hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

struct ProfScope {
  zk_ctx* c; int cls; int idx = -1;
  ProfScope(zk_ctx* c_, int cls_) : c(c_), cls(cls_) {
    if (!c->prof) return;
    idx = (int)c->ev_used.size();
    if (idx >= (int)c->ev_pool.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { idx = -1; return; }
      c->ev_pool.push_back({a, b});
    }
    c->ev_used.push_back({cls, idx});
    (void)hipEventRecord(c->ev_pool[idx].first, c->stream);
  }
  ~ProfScope() { if (idx >= 0) (void)hipEventRecord(c->ev_pool[idx].second, c->stream); }
};

// ---- fbank tables: mel_filter_bank(257,128,20,8000,16000,None,"kaldi",True), np.hanning(400), FFT twiddles ----
void build_tables(std::vector<double>& hann, std::vector<double>& tw, std::vector<double>& mel,
                  std::vector<int32_t>& lo, std::vector<int32_t>& hi) {
  const int M = ZK_FRAME_LEN;
  hann.resize(M);
  for (int i = 0; i < M; ++i) {
    const double n = (double)(1 - M + 2 * i);
    hann[i] = 0.5 + 0.5 * cos(M_PI * n / (double)(M - 1));
  }
  tw.resize(512);
  for (int k = 0; k < 256; ++k) {
    const double a = -2.0 * M_PI * (double)k / 512.0;
    tw[2 * k] = cos(a);
    tw[2 * k + 1] = sin(a);
  }
  auto h2m = [](double f) { return 1127.0 * log(1.0 + f / 700.0); };
  const int NF = ZK_NMEL, NB = ZK_NBINS;
  const double mel_min = h2m(20.0), mel_max = h2m(8000.0);
  std::vector<double> mf(NF + 2), ff(NB);
  const double step = (mel_max - mel_min) / (double)(NF + 1);
  for (int i = 0; i < NF + 2; ++i) mf[i] = (double)i * step + mel_min;
  mf[NF + 1] = mel_max;
  const double binw = 16000.0 / ((NB - 1) * 2.0);
  for (int k = 0; k < NB; ++k) ff[k] = h2m(binw * (double)k);
  mel.assign((size_t)NB * NF, 0.0);
  lo.assign(NF, NB);
  hi.assign(NF, 0);
  for (int k = 0; k < NB; ++k)
    for (int m = 0; m < NF; ++m) {
      const double down = -(mf[m] - ff[k]) / (mf[m + 1] - mf[m]);
      const double up = (mf[m + 2] - ff[k]) / (mf[m + 2] - mf[m + 1]);
      double v = down < up ? down : up;
      if (!(v > 0.0)) v = 0.0;
      mel[(size_t)k * NF + m] = v;
      if (v != 0.0) {
        if (k < lo[m]) lo[m] = k;
        if (k + 1 > hi[m]) hi[m] = k + 1;
      }
    }
  for (int m = 0; m < NF; ++m)
    if (hi[m] == 0) lo[m] = 0;  // all-zero filter: empty range
}

template <class T>
int upload(zk_ctx* c, const std::vector<T>& v, T** out) {
  HIPCHK(c, hipMalloc((void**)out, v.size() * sizeof(T)));
  HIPCHK(c, hipMemcpy(*out, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return ZK_OK;
}

int n_frames_for(int win) {
  if (win < ZK_FRAME_LEN) return 0;
  int n = 1 + (win - ZK_FRAME_LEN) / ZK_FRAME_HOP;
  return n > ZK_MAXLEN ? ZK_MAXLEN : n;
}

// ---- host tensor -> fp32 ----
float half_bits_to_float(uint16_t h) {
  const uint32_t s = (h >> 15) & 1, e = (h >> 10) & 0x1F, m = h & 0x3FF;
  uint32_t out;
  if (e == 0) {
    if (m == 0) out = s << 31;
    else {
      int ee = -1; uint32_t mm = m;
      do { ++ee; mm <<= 1; } while (!(mm & 0x400));
      out = (s << 31) | ((uint32_t)(127 - 15 - ee) << 23) | ((mm & 0x3FF) << 13);
    }
  } else if (e == 31) out = (s << 31) | 0x7F800000u | (m << 13);
  else out = (s << 31) | ((e - 15 + 127) << 23) | (m << 13);
  float f; memcpy(&f, &out, 4); return f;
}

// OCP fp8 e4m3 (bias 7, no infinities) -> float
float fp8_e4m3_to_float(uint8_t b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v;
  if (e == 0) v = ldexpf((float)m, -9);
  else if (e == 15 && m == 7) v = NAN;
  else v = ldexpf((float)(8 + m), e - 10);
  return s ? -v : v;
}

// value of a lo-plane entry: fp16 remainder, or byte 0 of the c8 pair (fp8 of the remainder times 2^11)
float lo_entry_to_float(uint16_t bits, int lo_fmt) {
  return lo_fmt == ZK_LO_C8 ? ldexpf(fp8_e4m3_to_float((uint8_t)(bits & 0xFF)), -ZK_C8_SHIFT) : half_bits_to_float(bits);
}

// byte 1 of a c8 pair must be the fp8 rounding of the full value (4 significant bits; subnormal step 2^-9; clamp 448)
bool c8_value_byte_ok(uint16_t bits, float v) {
  const float got = fp8_e4m3_to_float((uint8_t)(bits >> 8));
  const float want = fminf(fmaxf(v, -448.f), 448.f);
  return fabsf(got - want) <= fabsf(want) * 0.0626f + 0.001f;
}

bool to_f32(const zk_tensor_desc& t, std::vector<float>& out) {
  size_t n = 1;
  for (int i = 0; i < t.ndim; ++i) n *= (size_t)t.shape[i];
  out.resize(n);
  if (t.dtype == ZK_DT_F32) memcpy(out.data(), t.data, n * 4);
  else if (t.dtype == ZK_DT_F16) { const uint16_t* p = (const uint16_t*)t.data; for (size_t i = 0; i < n; ++i) out[i] = half_bits_to_float(p[i]); }
  else if (t.dtype == ZK_DT_BF16) { const uint16_t* p = (const uint16_t*)t.data; for (size_t i = 0; i < n; ++i) { uint32_t u = (uint32_t)p[i] << 16; memcpy(&out[i], &u, 4); } }
  else return false;
  return true;
}

struct TensorIndex {
  std::map<std::string, const zk_tensor_desc*> m;
  const zk_tensor_desc* find(std::initializer_list<std::string> names) const {
    for (auto& n : names) { auto it = m.find(n); if (it != m.end()) return it->second; }
    return nullptr;
  }
};

size_t numel(const zk_tensor_desc* t) { size_t n = 1; for (int i = 0; i < t->ndim; ++i) n *= (size_t)t->shape[i]; return n; }

int dev_f32(zk_ctx* c, StageModel& sm, const std::vector<float>& v, float** out) {
  HIPCHK(c, hipMalloc((void**)out, v.size() * 4));
  sm.allocs.push_back(*out);
  HIPCHK(c, hipMemcpyAsync(*out, v.data(), v.size() * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ZK_OK;
}

// exponent e of a weight tensor's c8 plane: the largest |w|·2^e lands in [112, 224] (fp8 e4m3 tops out at 448)
int c8_exponent(const float* w, size_t n) {
  float mx = 0.f;
  for (size_t i = 0; i < n; ++i) { const float a = fabsf(w[i]); if (a > mx && a < INFINITY) mx = a; }
  if (!(mx > 0.f)) return 0;
  int e = (int)floorf(log2f(224.0f / mx));
  if (e > 60) e = 60;
  if (e < -60) e = -60;
  return e;
}

int dev_planes(zk_ctx* c, StageModel& sm, const std::vector<float>& v, WMat* w) {
  HIPCHK(c, c->tmp_f32.ensure(v.size() * 4));
  HIPCHK(c, hipMemcpyAsync(c->tmp_f32.p, v.data(), v.size() * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMalloc((void**)&w->hi, v.size() * 2)); sm.allocs.push_back(w->hi);
  HIPCHK(c, hipMalloc((void**)&w->lo, v.size() * 2)); sm.allocs.push_back(w->lo);
  HIPCHK(c, hipMalloc((void**)&w->c8, v.size() * 2)); sm.allocs.push_back(w->c8);
  w->exp = c8_exponent(v.data(), v.size());
  zk_launch_split_c8(c->tmp_f32.as<float>(), (int64_t)v.size(), w->exp, 1, w->c8, c->stream);
  zk_launch_split_f32(c->tmp_f32.as<float>(), (int64_t)v.size(), 1.0f, w->hi, w->lo, c->stream);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ZK_OK;
}

int ensure_workspace(zk_ctx* c, int windows, bool split) {
  if (windows <= c->ws_windows && (!split || c->ws_split)) return ZK_OK;
  const int w = windows > c->ws_windows ? windows : c->ws_windows;
  const bool sp = split || c->ws_split;
  const size_t M = (size_t)w * ZK_SEQ + 256;
  HIPCHK(c, c->hidden.ensure(M * ZK_HIDDEN * 4));
  auto pl = [&](PlaneBuf& b, size_t elems) -> hipError_t {
    hipError_t e = b.hi.ensure(elems * 2);
    if (e != hipSuccess) return e;
    return sp ? b.lo.ensure(elems * 2) : hipSuccess;
  };
  HIPCHK(c, pl(c->patchA, M * ZK_PATCH_K));
  HIPCHK(c, pl(c->xn, M * ZK_HIDDEN));
  if (sp) HIPCHK(c, c->xn.rowexp.ensure(M * 4));
  HIPCHK(c, pl(c->qkv, M * 3 * ZK_HIDDEN));
  HIPCHK(c, pl(c->att, M * ZK_HIDDEN));
  HIPCHK(c, pl(c->mid, M * ZK_INTER));
  const size_t Ms = (size_t)w * 2 + 256;
  HIPCHK(c, c->hidden_s.ensure(Ms * ZK_HIDDEN * 4));
  HIPCHK(c, pl(c->att_s, Ms * ZK_HIDDEN));
  HIPCHK(c, pl(c->xn_s, Ms * ZK_HIDDEN));
  if (sp) HIPCHK(c, c->xn_s.rowexp.ensure(Ms * 4));
  if (sp) HIPCHK(c, c->xq_rowexp.ensure(((size_t)w * ZK_QROWS + 256) * 4));
  HIPCHK(c, pl(c->mid_s, Ms * ZK_INTER));
  c->ws_windows = w;
  c->ws_split = sp;
  return ZK_OK;
}

// direction of the next kernel of the chain (0 = first row first); flips with every call while walk_alt is on
int next_dir(zk_ctx* c) {
  if (!c->walk_alt) return 0;
  const int d = c->walk_dir;
  c->walk_dir ^= 1;
  return d;
}

int run_gemm(zk_ctx* c, int cls, zk_planes x, const WMat& w, const float* bias, int M, int N, int K, int epi, int nsplit,
              zk_planes out, float* resid, const float* pos, int lo_n_limit, int lo_c8_from = 1 << 30, int rev = 0,
              int lo_c8_to = 1 << 30, int patch_tr = 0, int ldo = 0) {
  ProfScope ps(c, cls);
  if (c->prof) c->prof_flops[cls] += 2.0 * M * (double)N * K;
  zk_gemm_args a;
  a.x_hi = x.hi; a.x_lo = x.lo; a.w_hi = w.hi; a.w_lo = (nsplit == ZK_F16C8) ? w.c8 : w.lo; a.bias = bias;
  a.x_rowexp = (nsplit == ZK_F16C8) ? x.rowexp : nullptr;
  a.M = M; a.N = N; a.K = K;
  a.o_hi = out.hi; a.o_lo = (nsplit != ZK_F16) ? out.lo : nullptr;
  a.resid = resid; a.pos = pos; a.lo_n_limit = lo_n_limit; a.lo_c8_from = lo_c8_from; a.lo_c8_to = lo_c8_to; a.w_exp = w.exp;
  a.rev = rev;
  a.patch_tr = patch_tr;
  a.ldo = ldo;
  a.x_tiled = (nsplit == ZK_F16C8) ? x.tiled : 0;
  a.o_tiled = (nsplit == ZK_F16C8 && epi == ZK_EPI_GELU) ? out.tiled : 0;
  a.x_rows = x.rows_cap / K;      // (PlaneBuf::get states the allocation in elements)
  if (nsplit == ZK_F16C8) {
    if (zk_launch_gemm_c8(a, epi, c->stream))
      return fail(c, ZK_E_SHAPE, "gemm %s: M=%d N=%d K=%d with x planes of %lld rows violates the launch contract (N%%256, K%%64, "
                  "x allocated for ceil(M/256)*256 rows)", kProfNames[cls], M, N, K, (long long)a.x_rows);
  } else zk_launch_gemm(a, epi, nsplit, c->stream);
  return ZK_OK;
}

// The patch embedding (0.4 % of a forward's time) runs as the 3-pass fp16 split even in ZK_F16C8: its operand is the
// INPUT, so its e4m3 correction error reaches every token of every layer — on the input-sensitive weight set it was the
// largest single GEMM term of the logit error (2.8e-4 of ~5e-4, tools/sens_budget.py), for free to remove.
#ifndef ZK_PATCH_X3
#define ZK_PATCH_X3 1
#endif
inline int patch_mode(int ns) { return (ZK_PATCH_X3 && ns == ZK_F16C8) ? ZK_F16X3 : ns; }
inline int patch_lo_fmt(int ns) { return patch_mode(ns) == ZK_F16C8 ? ZK_LO_C8 : ZK_LO_F16; }

#ifndef ZK_PRUNE_Q
#define ZK_PRUNE_Q 1        // 0: the pruned last layer still computes q for every token (A/B switch)
#endif
#ifndef ZK_MID_TILED
#define ZK_MID_TILED 1      // 0: row-major GELU planes between FC1 and FC2 (A/B switch)
#endif
#ifndef ZK_ATT_TILED
#define ZK_ATT_TILED 1      // 0: row-major attention output planes in front of the O projection (A/B switch)
#endif
#ifndef ZK_XN_TILED
#define ZK_XN_TILED 1       // 0: row-major LayerNorm planes in front of QKV / FC1 (A/B switch)
#endif
// forward of nb windows (one micro-batch) whose patch matrix is already in c->patchA.
// tr > 0: layer-0 constant-row reuse — patchA holds the real time patches only ([b][f][t < tr]); the constant rows of the
// residual stream and of the layer-0 q|k|v planes come from the model's table (build_l0_table), LayerNorm 1 and the QKV GEMM
// of layer 0 run on the 12·tr real rows of every window (1.5 % of a forward's FLOPs less; results bit-identical)
int forward_micro(zk_ctx* c, StageModel& sm, int nb, float* d_logits, int tr = 0) {
  const int M = nb * ZK_SEQ;
  float* hidden = c->hidden.as<float>();
  {
    const int ns = sm.base_mode();
    const bool sp = ns != ZK_F16;
    zk_planes pa = c->patchA.get(sp, patch_lo_fmt(ns));
  {
    ProfScope ps(c, P_EMBED);
    if (tr) zk_launch_l0_fill_hidden(hidden, sm.l0_hidden, nb, tr, c->stream);
    else zk_launch_cls_rows(hidden, sm.cls, sm.dist, sm.pos, nb, c->stream);
  }
  if (int grc = run_gemm(c, P_GEMM_PATCH, pa, sm.patch_w, sm.patch_b, nb * (tr ? ZK_FOUT * tr : ZK_NPATCH), ZK_HIDDEN, ZK_PATCH_K,
           ZK_EPI_PATCH, patch_mode(ns), zk_planes{nullptr, nullptr, 0}, hidden, sm.pos, 0, 1 << 30, 0, 1 << 30, tr)) return grc;
  }
  if (c->tap_layer == -1) {
    HIPCHK(c, c->tap.ensure((size_t)M * ZK_HIDDEN * 4));
    HIPCHK(c, hipMemcpyAsync(c->tap.p, hidden, (size_t)M * ZK_HIDDEN * 4, hipMemcpyDeviceToDevice, c->stream));
    c->tap_windows = nb;
  }
  bool pruned = false;
  c->walk_dir = 1;      // the patch GEMM above walked forwards: the first LayerNorm starts from the end
  for (int l = 0; l < sm.n_layers; ++l) {
    const LayerW& L = sm.L[l];
    // the layer's compute mode (ZK_F16MIX: per layer; otherwise the model's): the planes between its kernels follow it, the
    // fp32 residual stream is all that crosses a layer boundary
    const LayerMode lm = sm.layer_mode[l];
    // An operand's planes follow the GEMM that READS them: lo-plane format c8 byte pairs for a ZK_F16C8 consumer, else fp16
    // (of the QKV planes only k's columns are ever c8, for attention's fp8-corrected QK^T), and for a ZK_F16C8 consumer the
    // big operands that a GEMM-side kernel both writes and reads go as k-slice-major tiles (zk_planes::tiled; the
    // workspace planes hold whole 256-row blocks): LayerNorm -> QKV / FC1, FC1 -> FC2, attention -> O
    auto lof = [](int m) { return m == ZK_F16C8 ? ZK_LO_C8 : ZK_LO_F16; };
    const int ns_q = lm.qkv, ns_o = lm.o, ns_m = lm.mlp;
    const bool sp_q = ns_q != ZK_F16, sp_o = ns_o != ZK_F16, sp_m = ns_m != ZK_F16;
    zk_planes xn = c->xn.get(sp_q, lof(ns_q)), qkv = c->qkv.get(sp_q), att = c->att.get(sp_o, lof(ns_o)),
              xn2 = c->xn.get(sp_m, lof(ns_m)), mid = c->mid.get(sp_m, lof(ns_m));
    if (ns_q == ZK_F16C8) xn.tiled = ZK_XN_TILED;
    if (ns_o == ZK_F16C8) att.tiled = ZK_ATT_TILED;
    if (ns_m == ZK_F16C8) { xn2.tiled = ZK_XN_TILED; mid.tiled = ZK_MID_TILED; }
    // k's lo plane: c8 byte pairs exactly when attention runs the fp8-corrected QK^T
    const bool k_c8 = lm.att == ZK_F16C8;
    const int att_split = lm.att == ZK_F16C8 ? 2 : (lm.att == ZK_F16X3 ? 3 : 1);
    // profile classes of this layer's kernels (the ZK_F16X3 launches of a ZK_F16MIX model are timed apart)
    const bool mix = sm.mode == ZK_F16MIX;
    const int P_QKV = (mix && ns_q == ZK_F16X3) ? P_GEMM_QKV_X3 : P_GEMM_QKV, P_O = (mix && ns_o == ZK_F16X3) ? P_GEMM_O_X3 : P_GEMM_O,
              P_FC1 = (mix && ns_m == ZK_F16X3) ? P_GEMM_FC1_X3 : P_GEMM_FC1, P_FC2 = (mix && ns_m == ZK_F16X3) ? P_GEMM_FC2_X3 : P_GEMM_FC2,
              P_AT = (mix && lm.att == ZK_F16X3) ? P_ATTN_X3 : P_ATTN;
    // Only tokens 0/1 reach the head (ASTModel.forward:304), so in the LAST layer the queries, the attention output
    // projection and the MLP are needed for those two rows only (exact: same arithmetic per row).  K/V still need every
    // token.  Disabled while a debug tap wants the full residual stream of that layer.
    const bool last = (l == sm.n_layers - 1) && c->prune_last && c->tap_layer != l;
    // layer 0 with the constant-row table: LayerNorm 1 and the QKV GEMM see the real rows only (gathered into compact
    // planes), their q|k|v rows land in the MLP intermediate's buffer (idle until FC1) and one copy kernel assembles the
    // full planes from them and the table
    const bool l0 = tr > 0 && l == 0;
    const int Mq = l0 ? nb * ZK_FOUT * tr : M;
    zk_planes qkv_dst = qkv;
    if (l0) { qkv_dst = c->mid.get(sp_q); qkv_dst.lo_fmt = qkv.lo_fmt; }
    { ProfScope ps(c, P_LN); zk_launch_layernorm(hidden, ZK_HIDDEN, L.ln1_g, L.ln1_b, Mq, xn, sm.eps, c->stream, next_dir(c), l0 ? tr : 0); }
    // lo planes of the fused QKV: q fp16 (re-split by attention), k c8 byte pairs in ZK_F16C8 (fp8-corrected QK^T) else
    // fp16, v fp16 (attention's Vl·P pass)
    if (last && !l0 && ZK_PRUNE_Q) {
      // pruned last layer: k and v are needed for every token, q only for the tokens whose attention output is read —
      // tokens 0/1, computed together with the other 30 rows of their attention wave (ZK_QROWS: the wave's rescale decision
      // is wave-uniform, so the bits of rows 0/1 depend on what their wave-mates hold; with the whole wave's q exact the
      // logits equal the unpruned forward's bit for bit).  What the remaining rows of the query tile hold as "q" — the
      // previous layer's — only reaches output rows nobody reads.  The fused launch is cut to the k|v columns (a band of
      // the same planes: ldo = 2304); q comes from a 32-rows-per-window GEMM on gathered LayerNorm rows, staged in the MLP
      // intermediate's and the attention output's buffers (both idle at this point).
      WMat wkv = L.wqkv;
      wkv.hi += (size_t)ZK_HIDDEN * ZK_HIDDEN; wkv.lo += (size_t)ZK_HIDDEN * ZK_HIDDEN; wkv.c8 += (size_t)ZK_HIDDEN * ZK_HIDDEN;
      zk_planes kv = qkv;
      kv.hi += ZK_HIDDEN; if (kv.lo) kv.lo += ZK_HIDDEN;
      if (int grc = run_gemm(c, P_QKV, xn, wkv, L.bqkv + ZK_HIDDEN, M, 2 * ZK_HIDDEN, ZK_HIDDEN, ZK_EPI_STORE, ns_q, kv,
               nullptr, nullptr, 2 * ZK_HIDDEN, k_c8 ? 0 : 1 << 30, next_dir(c), ZK_HIDDEN, 0, 3 * ZK_HIDDEN)) return grc;
      zk_planes xq = c->mid.get(sp_q, lof(ns_q)), q32 = c->att.get(sp_q);
      xq.rowexp = (sp_q && ns_q == ZK_F16C8) ? c->xq_rowexp.as<int32_t>() : nullptr;
      { ProfScope ps(c, P_EMBED); zk_launch_gather_xq(xn, nb, xq, c->stream); }
      if (int grc = run_gemm(c, P_QKV, xq, L.wqkv, L.bqkv, ZK_QROWS * nb, ZK_HIDDEN, ZK_HIDDEN, ZK_EPI_STORE, ns_q, q32,
               nullptr, nullptr, ZK_HIDDEN)) return grc;
      { ProfScope ps(c, P_EMBED); zk_launch_scatter_q(q32, nb, qkv, c->stream); }
    } else
    if (int grc = run_gemm(c, P_QKV, xn, L.wqkv, L.bqkv, Mq, 3 * ZK_HIDDEN, ZK_HIDDEN, ZK_EPI_STORE, ns_q, qkv_dst,
             nullptr, nullptr, 3 * ZK_HIDDEN, k_c8 ? ZK_HIDDEN : 1 << 30, next_dir(c), 2 * ZK_HIDDEN)) return grc;
    // layer-0 constant-row attention (zk_common.h): the constant queries continue from the model's tabulated state over the
    // constant keys and only see the window's real keys; the real queries see all keys in the order [constant | real].  The
    // full q|k|v planes are never assembled: the window's 128-row tail planes (in the q|k|v workspace) are all it adds.
    const bool l0att = l0 && !last && sm.l0_att && c->l0_attention && sp_q && att_split != 1;
    if (l0att) {
      zk_planes ctab{sm.l0_ctab_hi, sm.l0_ctab_lo, ZK_LO_F16}, tail = qkv;
      { ProfScope ps(c, P_EMBED); zk_launch_l0_tail(qkv_dst, ctab, tail, nb, tr, c->stream); }
      ProfScope ps(c, P_AT);
      const int n_real = ZK_FOUT * tr, n_const = ZK_SEQ - n_real;
      if (c->prof) c->prof_flops[P_AT] += (double)nb * ZK_HEADS * 4.0 * ZK_HEAD_DIM *
                                          ((double)n_const * (ZK_SEQ - 1024) + (double)n_real * ZK_SEQ);
      (void)next_dir(c);
      zk_launch_attention_l0(ZK_L0_ATT_CONST, ctab, sm.l0_state, tail, att, nb, tr, att_split, c->stream);
      zk_launch_attention_l0(ZK_L0_ATT_REAL, ctab, sm.l0_state, tail, att, nb, tr, att_split, c->stream);
    } else {
    if (l0) {
      ProfScope ps(c, P_EMBED);
      zk_planes tab{sm.l0_qkv_hi, sp_q ? sm.l0_qkv_lo : nullptr, qkv.lo_fmt};
      zk_launch_l0_assemble_qkv(qkv_dst, tab, qkv, nb, tr, c->stream);
    }
    {
      ProfScope ps(c, P_AT);
      // pruned last layer: only the query rows whose q exists are worth computing — one wave's 32 with ZK_PRUNE_Q (the
      // other waves of the workgroup still help staging K / V), else the first 128-row block
      const bool q32 = last && l > 0 && ZK_PRUNE_Q == 1;
      const int qt = last ? (q32 ? -ZK_QROWS : 1) : 10;
      if (c->prof) c->prof_flops[P_AT] += (double)nb * ZK_HEADS * 4.0 * (last ? (q32 ? (double)ZK_QROWS : 128.0) : (double)ZK_SEQ) * ZK_SEQ * ZK_HEAD_DIM;
      zk_launch_attention(qkv, att, nb, att_split, qt, c->stream, next_dir(c));
    }
    }      // (!l0att)
    if (last) {
      zk_planes att_s = c->att_s.get(sp_o, lof(ns_o)), xn_s = c->xn_s.get(sp_m, lof(ns_m)), mid_s = c->mid_s.get(sp_m, lof(ns_m));
      float* hs = c->hidden_s.as<float>();
      { ProfScope ps(c, P_EMBED); zk_launch_gather_tok01(att, hidden, nb, att_s, hs, c->stream); }
      if (int grc = run_gemm(c, P_O, att_s, L.wo, L.bo, 2 * nb, ZK_HIDDEN, ZK_HIDDEN, ZK_EPI_RESID, ns_o,
               zk_planes{nullptr, nullptr, 0}, hs, nullptr, 0)) return grc;
      { ProfScope ps(c, P_LN); zk_launch_layernorm(hs, ZK_HIDDEN, L.ln2_g, L.ln2_b, 2 * nb, xn_s, sm.eps, c->stream); }
      if (int grc = run_gemm(c, P_FC1, xn_s, L.w1, L.b1, 2 * nb, ZK_INTER, ZK_HIDDEN, ZK_EPI_GELU, ns_m, mid_s, nullptr,
               nullptr, ZK_INTER)) return grc;
      if (int grc = run_gemm(c, P_FC2, mid_s, L.w2, L.b2, 2 * nb, ZK_HIDDEN, ZK_INTER, ZK_EPI_RESID, ns_m,
               zk_planes{nullptr, nullptr, 0}, hs, nullptr, 0)) return grc;
      pruned = true;
      continue;
    }
    if (int grc = run_gemm(c, P_O, att, L.wo, L.bo, M, ZK_HIDDEN, ZK_HIDDEN, ZK_EPI_RESID, ns_o,
             zk_planes{nullptr, nullptr, 0}, hidden, nullptr, 0, 1 << 30, next_dir(c))) return grc;
    { ProfScope ps(c, P_LN); zk_launch_layernorm(hidden, ZK_HIDDEN, L.ln2_g, L.ln2_b, M, xn2, sm.eps, c->stream, next_dir(c)); }
    if (int grc = run_gemm(c, P_FC1, xn2, L.w1, L.b1, M, ZK_INTER, ZK_HIDDEN, ZK_EPI_GELU, ns_m, mid, nullptr, nullptr,
             ZK_INTER, 1 << 30, next_dir(c))) return grc;
    if (int grc = run_gemm(c, P_FC2, mid, L.w2, L.b2, M, ZK_HIDDEN, ZK_INTER, ZK_EPI_RESID, ns_m,
             zk_planes{nullptr, nullptr, 0}, hidden, nullptr, 0, 1 << 30, next_dir(c))) return grc;
    if (c->tap_layer == l) {
      HIPCHK(c, c->tap.ensure((size_t)M * ZK_HIDDEN * 4));
      HIPCHK(c, hipMemcpyAsync(c->tap.p, hidden, (size_t)M * ZK_HIDDEN * 4, hipMemcpyDeviceToDevice, c->stream));
      c->tap_windows = nb;
    }
  }
  {
    ProfScope ps(c, P_HEAD);
    zk_launch_head(pruned ? c->hidden_s.as<float>() : hidden, pruned ? 2 : ZK_SEQ, nb, sm.lnf_g, sm.lnf_b, sm.lnh_g,
                   sm.lnh_b, sm.head_w, sm.head_b, sm.num_labels, sm.eps, d_logits, c->stream);
  }
  HIPCHK(c, hipGetLastError());
  return ZK_OK;
}

// Layer-0 constant-row table of a stage model.  A 1 s window fills 98 of the extractor's 1024 frames, so of its 1212 patch
// tokens only those with t < t_real = ceil(n_frames / 10) (120 of them) see real frames; the input of layer 0 for every
// other row — cls, distillation, the 1092 padding-only patches — is the same for every window: pad value · patch filters +
// bias + position.  Their residual rows and their layer-0 LayerNorm + q|k|v rows are therefore computed ONCE, by the
// ordinary kernels on one window (every kernel involved is row-wise, so a row's bits do not depend on what else is in the
// launch: the micro-batch invariance tests pin that), and copied from then on.  Depends on the weights, the compute
// mode, the extractor's mean / std (the pad value) and n_frames.
int build_l0_table(zk_ctx* c, StageModel& sm, int n_frames) {
  const int ns = sm.layer_mode[0].qkv, nsb = sm.base_mode();
  const bool k_c8 = sm.layer_mode[0].att == ZK_F16C8;      // k's lo plane as c8 byte pairs (attention's fp8-corrected QK^T)
  const bool sp = ns != ZK_F16, spb = nsb != ZK_F16;
  const int lf = (ns == ZK_F16C8) ? ZK_LO_C8 : ZK_LO_F16;
  sm.l0_frames = -1;      // until the table below is complete the forward takes the ordinary path
  int rc = ensure_workspace(c, 1, sp || spb);
  if (rc) return rc;
  if (!sm.l0_hidden || !sm.l0_qkv_hi || !sm.l0_qkv_lo) {      // all three or none: a failed hipMalloc leaves the model without a table
    void* t[3] = {nullptr, nullptr, nullptr};
    const size_t bytes[3] = {(size_t)ZK_SEQ * ZK_HIDDEN * 4, (size_t)ZK_SEQ * 3 * ZK_HIDDEN * 2, (size_t)ZK_SEQ * 3 * ZK_HIDDEN * 2};
    for (int i = 0; i < 3; ++i)
      if (hipMalloc(&t[i], bytes[i]) != hipSuccess) {
        (void)hipGetLastError();
        for (int j = 0; j < i; ++j) (void)hipFree(t[j]);
        return fail(c, ZK_E_NOMEM, "layer-0 constant-row table: out of device memory");
      }
    sm.l0_hidden = (float*)t[0]; sm.l0_qkv_hi = (half_t*)t[1]; sm.l0_qkv_lo = (half_t*)t[2];
    for (void* p : t) sm.allocs.push_back(p);
  }
  const bool was_prof = c->prof;
  c->prof = false;      // a one-off: keep it out of the per-class timings of a profiled region
  float* hidden = c->hidden.as<float>();
  zk_planes pa = c->patchA.get(spb, patch_lo_fmt(nsb)), xn = c->xn.get(sp, lf), qkv = c->qkv.get(sp);
  if (ns == ZK_F16C8) xn.tiled = ZK_XN_TILED;
  const LayerW& L = sm.L[0];
  // window 0 of the feature slot: which window it is does not matter for the rows that are kept
  zk_launch_im2col_compact(c->feat.as<float>(), n_frames, nullptr, 1, sm.mean, sm.std * 2.0f, pa, c->stream);
  zk_launch_cls_rows(hidden, sm.cls, sm.dist, sm.pos, 1, c->stream);
  rc = run_gemm(c, P_GEMM_PATCH, pa, sm.patch_w, sm.patch_b, ZK_NPATCH, ZK_HIDDEN, ZK_PATCH_K, ZK_EPI_PATCH, patch_mode(nsb),
                zk_planes{nullptr, nullptr, 0}, hidden, sm.pos, 0);
  if (!rc) {
    zk_launch_layernorm(hidden, ZK_HIDDEN, L.ln1_g, L.ln1_b, ZK_SEQ, xn, sm.eps, c->stream);
    rc = run_gemm(c, P_GEMM_QKV, xn, L.wqkv, L.bqkv, ZK_SEQ, 3 * ZK_HIDDEN, ZK_HIDDEN, ZK_EPI_STORE, ns, qkv, nullptr, nullptr,
                  3 * ZK_HIDDEN, k_c8 ? ZK_HIDDEN : 1 << 30, 0, 2 * ZK_HIDDEN);
  }
  c->prof = was_prof;
  if (rc) return rc;
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(sm.l0_hidden, hidden, (size_t)ZK_SEQ * ZK_HIDDEN * 4, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(sm.l0_qkv_hi, qkv.hi, (size_t)ZK_SEQ * 3 * ZK_HIDDEN * 2, hipMemcpyDeviceToDevice, c->stream));
  if (sp) HIPCHK(c, hipMemcpyAsync(sm.l0_qkv_lo, qkv.lo, (size_t)ZK_SEQ * 3 * ZK_HIDDEN * 2, hipMemcpyDeviceToDevice, c->stream));
  // a one-off per model: wait for the copies, so that the table is complete whatever stream a later forward runs on
  // (zk_set_stream, async mode)
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // constant-row attention: the table once more in constant-token order, and the state of the constant queries over the
  // constant keys 0..1023 (computed by the attention kernel itself, so a window's launch continues exactly where it stops)
  sm.l0_att = false;
  const int trr = (n_frames + ZK_TSTRIDE - 1) / ZK_TSTRIDE;
  const int am = sm.layer_mode[0].att;
  if (sp && am != ZK_F16 && zk_l0_att_supported(trr)) {
    if (!sm.l0_ctab_hi || !sm.l0_ctab_lo || !sm.l0_state) {
      void* t[3] = {nullptr, nullptr, nullptr};
      const size_t bytes[3] = {(size_t)ZK_L0_CTAB_ROWS * 3 * ZK_HIDDEN * 2, (size_t)ZK_L0_CTAB_ROWS * 3 * ZK_HIDDEN * 2,
                               (size_t)ZK_HEADS * ZK_SEQ * ZK_L0_STATE_LD * 4};
      bool ok = true;
      for (int i = 0; i < 3 && ok; ++i)
        if (hipMalloc(&t[i], bytes[i]) != hipSuccess) { (void)hipGetLastError(); for (int j = 0; j < i; ++j) (void)hipFree(t[j]); ok = false; }
      if (ok) {
        sm.l0_ctab_hi = (half_t*)t[0]; sm.l0_ctab_lo = (half_t*)t[1]; sm.l0_state = (float*)t[2];
        for (void* p : t) sm.allocs.push_back(p);
      }
    }
    if (sm.l0_ctab_hi && sm.l0_ctab_lo && sm.l0_state) {
      zk_planes tabn{sm.l0_qkv_hi, sm.l0_qkv_lo, ZK_LO_F16}, ctab{sm.l0_ctab_hi, sm.l0_ctab_lo, ZK_LO_F16};
      zk_launch_l0_const_order(tabn, ctab, trr, ZK_L0_CTAB_ROWS, c->stream);
      zk_launch_attention_l0(ZK_L0_ATT_DUMP, ctab, sm.l0_state, zk_planes{nullptr, nullptr, 0}, zk_planes{nullptr, nullptr, 0}, 1, trr,
                             am == ZK_F16C8 ? 2 : 3, c->stream);
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipStreamSynchronize(c->stream));
      sm.l0_att = true;
    }
  }
  sm.l0_frames = n_frames; sm.l0_mode = (ns * 2 + (k_c8 ? 1 : 0)) * 4 + am; sm.l0_base = nsb; sm.l0_mean = sm.mean; sm.l0_std = sm.std;
  return ZK_OK;
}

// src_full != nullptr: device (B,1024,128) normalised; else feature slot with optional device index list
int forward_device(zk_ctx* c, int stage, const float* src_full, const int32_t* d_idx, int B, float* d_logits) {
  StageModel& sm = c->model[stage];
  const bool sp = sm.any_split();
  const bool spb = sm.base_mode() != ZK_F16;      // the patch matrix follows the model's base mode
  // micro_batch == 0 (auto).  Large batches: as few, equal micro-batches of at most 512 windows as possible (≈ 21 GB of
  // activations; every GEMM then runs >= 12 rounds of the 256 persistent workgroups and the per-launch tails and
  // the ragged last micro-batch stop mattering: 1024 windows as 2 x 512 measured +3 % over 9 x 107 + 61).  Small
  // batches: the size whose 256-row tile count fills the 256 workgroups with the least round-up waste for the
  // N=768 GEMMs (3 column tiles): tiles_m*3 just below a multiple of 256.
  int mbs = c->micro_batch;
  if (mbs <= 0) {
    if (B >= 214) {
      const int n = (B + 511) / 512;
      mbs = (B + n - 1) / n;
    } else {
      static const int good[] = {107, 89, 71, 53, 35, 17};
      mbs = 17;
      for (int g : good) if (B >= g) { mbs = g; break; }
      if (B < 17) mbs = B;
    }
  }
  int rc = ensure_workspace(c, B < mbs ? B : mbs, sp);
  if (rc) return rc;
  // layer-0 constant-row reuse: only for the feature slot (a caller's full (B,1024,128) input may hold anything in its
  // padding rows) and only when some time patches are padding-only
  int tr = 0;
  if (!src_full && c->l0_reuse && c->feat_frames > 0) {
    tr = (c->feat_frames + ZK_TSTRIDE - 1) / ZK_TSTRIDE;
    if (tr >= ZK_TOUT) tr = 0;
  }
  if (tr && (sm.l0_frames != c->feat_frames ||
             sm.l0_mode != (sm.layer_mode[0].qkv * 2 + (sm.layer_mode[0].att == ZK_F16C8 ? 1 : 0)) * 4 + sm.layer_mode[0].att ||
             sm.l0_base != sm.base_mode() ||
             sm.l0_mean != sm.mean || sm.l0_std != sm.std)) {
    // a table that cannot be built (out of memory) only costs the shortcut: the forward takes the ordinary path
    if (build_l0_table(c, sm, c->feat_frames) != ZK_OK) tr = 0;
  }
  bool tapped = false;
  const int saved_tap = c->tap_layer;
  for (int b0 = 0; b0 < B; b0 += mbs) {
    const int nb = (B - b0) < mbs ? (B - b0) : mbs;
    {
      ProfScope ps(c, P_EMBED);
      if (src_full)
        zk_launch_im2col_full(src_full + (size_t)b0 * ZK_MAXLEN * ZK_NMEL, nb, c->patchA.get(spb, patch_lo_fmt(sm.base_mode())), c->stream);
      else
        zk_launch_im2col_compact(c->feat.as<float>() + (d_idx ? 0 : (size_t)b0 * c->feat_frames * ZK_NMEL),
                                 c->feat_frames, d_idx ? d_idx + b0 : nullptr, nb, sm.mean, sm.std * 2.0f,
                                 c->patchA.get(spb, patch_lo_fmt(sm.base_mode())), c->stream, tr);
    }
    if (tapped) c->tap_layer = -2;  // tap only the first micro-batch
    rc = forward_micro(c, sm, nb, d_logits + (size_t)b0 * sm.num_labels, tr);
    tapped = true;
    if (rc) { c->tap_layer = saved_tap; return rc; }
  }
  c->tap_layer = saved_tap;
  return ZK_OK;
}

int finish(zk_ctx* c) {
  if (!c->async) HIPCHK(c, hipStreamSynchronize(c->stream));
  return ZK_OK;
}

// copy helpers honouring host|device on either side
int to_device(zk_ctx* c, const void* src, size_t bytes, DevBuf& stage, const void** out) {
  if (is_device_ptr(src)) { *out = src; return ZK_OK; }
  HIPCHK(c, stage.ensure(bytes));
  HIPCHK(c, hipMemcpyAsync(stage.p, src, bytes, hipMemcpyHostToDevice, c->stream));
  *out = stage.p;
  return ZK_OK;
}

int from_device(zk_ctx* c, const void* dsrc, void* dst, size_t bytes) {
  if (bytes == 0) return ZK_OK;
  const hipMemcpyKind k = is_device_ptr(dst) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  HIPCHK(c, hipMemcpyAsync(dst, dsrc, bytes, k, c->stream));
  if (k == hipMemcpyDeviceToHost) HIPCHK(c, hipStreamSynchronize(c->stream));
  return ZK_OK;
}

int check_stage(zk_ctx* c, int stage) {
  if (!c) return ZK_E_ARG;
  if (stage < 0 || stage > 1) return fail(c, ZK_E_ARG, "stage must be 0 or 1, got %d", stage);
  if (!c->model[stage].loaded) return fail(c, ZK_E_STATE, "stage %d has no model loaded (zk_model_load)", stage);
  return ZK_OK;
}

}  // namespace

// ---- context accessors for comm.hip --------------------------------------------------------------------------------
hipStream_t zk_ctx_stream(zk_ctx* c) { return c->stream; }
int zk_ctx_device(zk_ctx* c) { return c->device; }
int zk_ctx_fail(zk_ctx* c, int code, const char* msg) { return fail(c, code, "%s", msg); }
void** zk_ctx_comm_slot(zk_ctx* c) { return &c->comm; }
// HIP-event bracket of the "allgather" profile class around what comm.hip queues on the context's stream
void* zk_ctx_prof_open_allgather(zk_ctx* c) { return c->prof ? new ProfScope(c, P_ALLGATHER) : nullptr; }
void zk_ctx_prof_close(void* scope) { delete (ProfScope*)scope; }
void* zk_ctx_stage_buf(zk_ctx* c, int which, size_t bytes) {
  if (c->comm_stage[which & 1].ensure(bytes) != hipSuccess) { fail(c, ZK_E_NOMEM, "comm staging buffer of %zu bytes", bytes); return nullptr; }
  return c->comm_stage[which & 1].p;
}

// =====================================================================================================================
extern "C" {

const char* zk_version(void) { return "zkast 0.5 (gfx950)"; }

int zk_create(int device_id, zk_ctx** out) {
  if (!out) return fail(nullptr, ZK_E_ARG, "out is NULL");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return fail(nullptr, ZK_E_HIP, "no HIP device visible: libzkast needs an MI355X (gfx950) GPU; there is no CPU fallback");
  }
  if (device_id < 0 || device_id >= n) return fail(nullptr, ZK_E_ARG, "device_id %d out of range [0,%d)", device_id, n);
  zk_ctx* c = new zk_ctx();
  if (const char* wa = getenv("ZK_WALK_ALT")) c->walk_alt = wa[0] == '1';
  if (const char* lr = getenv("ZK_L0_REUSE")) c->l0_reuse = lr[0] != '0';      // A/B switch; zk_set_layer0_reuse is the API
  if (const char* la = getenv("ZK_L0_ATT")) c->l0_attention = la[0] != '0';    // A/B switch; zk_set_layer0_attention is the API
  c->device = device_id;
  if (hipSetDevice(device_id) != hipSuccess) { delete c; return fail(nullptr, ZK_E_HIP, "hipSetDevice(%d) failed", device_id); }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) {
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
      std::string a = prop.gcnArchName;
      delete c;
      return fail(nullptr, ZK_E_HIP, "device %d is %s; libzkast is built for gfx950 only", device_id, a.c_str());
    }
  }
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return fail(nullptr, ZK_E_HIP, "hipStreamCreate failed"); }
  c->stream = c->own_stream;
  std::vector<double> hann, tw, mel;
  std::vector<int32_t> lo, hi;
  build_tables(hann, tw, mel, lo, hi);
  {
    int band = 0;
    for (int m = 0; m < ZK_NMEL; ++m) band += hi[m] - lo[m];
    if (band > ZK_MEL_BAND_MAX) { zk_destroy(c); return fail(nullptr, ZK_E_STATE, "mel band of %d entries exceeds the log-mel kernel's LDS table (%d)", band, ZK_MEL_BAND_MAX); }
  }
  int rc = upload(c, hann, &c->d_hann);
  if (!rc) rc = upload(c, tw, &c->d_tw);
  if (!rc) rc = upload(c, mel, &c->d_mel);
  if (!rc) rc = upload(c, lo, &c->d_mel_lo);
  if (!rc) rc = upload(c, hi, &c->d_mel_hi);
  if (rc) { g_create_err = c->err; zk_destroy(c); return rc; }
  *out = c;
  return ZK_OK;
}

void zk_destroy(zk_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  (void)zk_comm_destroy(c);
  c->comm_stage[0].release(); c->comm_stage[1].release();
  for (auto& m : c->model) m.release();
  for (void* p : {(void*)c->d_hann, (void*)c->d_tw, (void*)c->d_mel, (void*)c->d_mel_lo, (void*)c->d_mel_hi})
    if (p) (void)hipFree(p);
  c->audio_slot.release();
  for (DevBuf* b : {&c->feat, &c->st_in, &c->st_out, &c->st_idx, &c->audio_dev, &c->s1_logits, &c->s2_logits, &c->gate_idx,
                    &c->gate_cnt, &c->tmp_f32, &c->hidden, &c->rs_kern, &c->tap})
    b->release();
  c->hidden_s.release();
  c->xq_rowexp.release();
  for (PlaneBuf* b : {&c->patchA, &c->xn, &c->qkv, &c->att, &c->mid, &c->att_s, &c->xn_s, &c->mid_s}) { b->hi.release(); b->lo.release(); b->rowexp.release(); }
  for (auto& e : c->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

const char* zk_last_error(zk_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

int zk_set_stream(zk_ctx* c, void* s) {
  if (!c) return ZK_E_ARG;
  c->stream = s ? (hipStream_t)s : c->own_stream;
  return ZK_OK;
}
int zk_set_async(zk_ctx* c, int e) { if (!c) return ZK_E_ARG; c->async = e != 0; return ZK_OK; }
int zk_synchronize(zk_ctx* c) { if (!c) return ZK_E_ARG; HIPCHK(c, hipStreamSynchronize(c->stream)); return ZK_OK; }
int zk_set_layer0_attention(zk_ctx* c, int e) { if (!c) return ZK_E_ARG; c->l0_attention = e != 0; return ZK_OK; }
int zk_set_micro_batch(zk_ctx* c, int32_t w) {
  if (!c) return ZK_E_ARG;
  // 512 windows (M = 621,568 token rows, ~21 GB of activation planes) is the largest micro-batch the test suite runs
  if (w < 0 || w > 512) return fail(c, ZK_E_ARG, "micro batch %d out of range [0,512] (0 = auto)", w);
  c->micro_batch = w;
  return ZK_OK;
}

int zk_model_load(zk_ctx* c, int stage, const zk_tensor_desc* t, int32_t n, const zk_ast_config* cfg, float fx_mean,
                  float fx_std, int32_t mode) {
  if (!c) return ZK_E_ARG;
  if (stage < 0 || stage > 1) return fail(c, ZK_E_ARG, "stage must be 0 or 1");
  if (!t || n <= 0 || !cfg) return fail(c, ZK_E_ARG, "tensors/cfg missing");
  if (mode != ZK_F16 && mode != ZK_F16X3 && mode != ZK_F16C8 && mode != ZK_F16MIX)
    return fail(c, ZK_E_ARG, "compute_mode must be ZK_F16 (1), ZK_F16C8 (2), ZK_F16X3 (3) or ZK_F16MIX (4)");
  if (cfg->hidden_size != ZK_HIDDEN || cfg->num_attention_heads != ZK_HEADS || cfg->intermediate_size != ZK_INTER ||
      cfg->patch_size != ZK_PATCH || cfg->frequency_stride != ZK_FSTRIDE || cfg->time_stride != ZK_TSTRIDE ||
      cfg->max_length != ZK_MAXLEN || cfg->num_mel_bins != ZK_NMEL)
    return fail(c, ZK_E_SHAPE, "unsupported ASTConfig: this build is specialised to hidden 768 / 12 heads / mlp 3072 / "
                               "patch 16 stride 10x10 / 1024x128 input");
  if (cfg->num_hidden_layers < 1 || cfg->num_hidden_layers > ZK_LAYERS) return fail(c, ZK_E_SHAPE, "num_hidden_layers must be 1..12");
  if (cfg->num_labels < 1 || cfg->num_labels > 64) return fail(c, ZK_E_SHAPE, "num_labels must be 1..64");
  if (!(fx_std > 0.f)) return fail(c, ZK_E_ARG, "fx_std must be > 0");
  HIPCHK(c, hipSetDevice(c->device));
  StageModel& sm = c->model[stage];
  sm.release();
  sm.set_mode(mode); sm.num_labels = cfg->num_labels; sm.n_layers = cfg->num_hidden_layers; sm.eps = cfg->layer_norm_eps;
  sm.mean = fx_mean; sm.std = fx_std;

  TensorIndex ix;
  for (int i = 0; i < n; ++i) if (t[i].name && t[i].data) ix.m[t[i].name] = &t[i];
  const std::string P = "audio_spectrogram_transformer.";
  std::vector<float> v, v2;
  auto need = [&](std::initializer_list<std::string> names, size_t want, std::vector<float>& out) -> int {
    const zk_tensor_desc* d = ix.find(names);
    if (!d) return fail(c, ZK_E_SHAPE, "missing tensor '%s'", names.begin()->c_str());
    if (numel(d) != want) return fail(c, ZK_E_SHAPE, "tensor '%s' has %zu elements, expected %zu", d->name, numel(d), want);
    if (!to_f32(*d, out)) return fail(c, ZK_E_ARG, "tensor '%s': unsupported dtype %d", d->name, d->dtype);
    return ZK_OK;
  };
  int rc;
#define NEEDF(dst, want, ...) do { if ((rc = need({__VA_ARGS__}, (want), v))) return rc; if ((rc = dev_f32(c, sm, v, &(dst)))) return rc; } while (0)
#define NEEDP(w, want, ...) do { if ((rc = need({__VA_ARGS__}, (want), v))) return rc; if ((rc = dev_planes(c, sm, v, &(w)))) return rc; } while (0)
  NEEDF(sm.cls, ZK_HIDDEN, P + "embeddings.cls_token");
  NEEDF(sm.dist, ZK_HIDDEN, P + "embeddings.distillation_token");
  NEEDF(sm.pos, (size_t)ZK_SEQ * ZK_HIDDEN, P + "embeddings.position_embeddings");
  NEEDF(sm.patch_b, ZK_HIDDEN, P + "embeddings.patch_embeddings.projection.bias");
  NEEDP(sm.patch_w, (size_t)ZK_HIDDEN * ZK_PATCH_K, P + "embeddings.patch_embeddings.projection.weight");
  for (int l = 0; l < sm.n_layers; ++l) {
    LayerW& L = sm.L[l];
    const std::string a5 = P + "layers." + std::to_string(l) + ".";
    const std::string a4 = P + "encoder.layer." + std::to_string(l) + ".";
    // fused QKV weight [2304,768] and bias
    std::vector<float> wq((size_t)3 * ZK_HIDDEN * ZK_HIDDEN), bq(3 * ZK_HIDDEN);
    const char* n5[3] = {"attention.q_proj", "attention.k_proj", "attention.v_proj"};
    const char* n4[3] = {"attention.attention.query", "attention.attention.key", "attention.attention.value"};
    for (int j = 0; j < 3; ++j) {
      if ((rc = need({a5 + n5[j] + ".weight", a4 + n4[j] + ".weight"}, (size_t)ZK_HIDDEN * ZK_HIDDEN, v))) return rc;
      memcpy(wq.data() + (size_t)j * ZK_HIDDEN * ZK_HIDDEN, v.data(), v.size() * 4);
      if ((rc = need({a5 + n5[j] + ".bias", a4 + n4[j] + ".bias"}, ZK_HIDDEN, v))) return rc;
      memcpy(bq.data() + (size_t)j * ZK_HIDDEN, v.data(), v.size() * 4);
    }
    if ((rc = dev_planes(c, sm, wq, &L.wqkv))) return rc;
    if ((rc = dev_f32(c, sm, bq, &L.bqkv))) return rc;
    NEEDP(L.wo, (size_t)ZK_HIDDEN * ZK_HIDDEN, a5 + "attention.o_proj.weight", a4 + "attention.output.dense.weight");
    NEEDF(L.bo, ZK_HIDDEN, a5 + "attention.o_proj.bias", a4 + "attention.output.dense.bias");
    NEEDP(L.w1, (size_t)ZK_INTER * ZK_HIDDEN, a5 + "mlp.fc1.weight", a4 + "intermediate.dense.weight");
    NEEDF(L.b1, ZK_INTER, a5 + "mlp.fc1.bias", a4 + "intermediate.dense.bias");
    NEEDP(L.w2, (size_t)ZK_INTER * ZK_HIDDEN, a5 + "mlp.fc2.weight", a4 + "output.dense.weight");
    NEEDF(L.b2, ZK_HIDDEN, a5 + "mlp.fc2.bias", a4 + "output.dense.bias");
    NEEDF(L.ln1_g, ZK_HIDDEN, a5 + "layernorm_before.weight", a4 + "layernorm_before.weight");
    NEEDF(L.ln1_b, ZK_HIDDEN, a5 + "layernorm_before.bias", a4 + "layernorm_before.bias");
    NEEDF(L.ln2_g, ZK_HIDDEN, a5 + "layernorm_after.weight", a4 + "layernorm_after.weight");
    NEEDF(L.ln2_b, ZK_HIDDEN, a5 + "layernorm_after.bias", a4 + "layernorm_after.bias");
  }
  NEEDF(sm.lnf_g, ZK_HIDDEN, P + "layernorm.weight");
  NEEDF(sm.lnf_b, ZK_HIDDEN, P + "layernorm.bias");
  NEEDF(sm.lnh_g, ZK_HIDDEN, "classifier.layernorm.weight");
  NEEDF(sm.lnh_b, ZK_HIDDEN, "classifier.layernorm.bias");
  NEEDF(sm.head_w, (size_t)sm.num_labels * ZK_HIDDEN, "classifier.dense.weight");
  NEEDF(sm.head_b, (size_t)sm.num_labels, "classifier.dense.bias");
#undef NEEDF
#undef NEEDP
  sm.loaded = true;
  return ZK_OK;
}

int zk_model_set_compute_mode(zk_ctx* c, int stage, int32_t mode) {
  int rc = check_stage(c, stage);
  if (rc) return rc;
  if (mode != ZK_F16 && mode != ZK_F16X3 && mode != ZK_F16C8 && mode != ZK_F16MIX) return fail(c, ZK_E_ARG, "compute_mode must be 1, 2, 3 or 4");
  c->model[stage].set_mode(mode);
  return ZK_OK;
}

int zk_model_set_layer_modes(zk_ctx* c, int stage, const int32_t* modes, int32_t n) {
  int rc = check_stage(c, stage);
  if (rc) return rc;
  StageModel& sm = c->model[stage];
  if (!modes || (n != sm.n_layers && n != 4 * sm.n_layers))
    return fail(c, ZK_E_ARG, "zk_model_set_layer_modes: need %d modes (one per encoder layer) or %d (QKV, QK^T, O, MLP of each layer), got %d",
                sm.n_layers, 4 * sm.n_layers, n);
  const bool per_kind = n == 4 * sm.n_layers;
  LayerMode lm[ZK_LAYERS];
  for (int l = 0; l < sm.n_layers; ++l) {
    lm[l] = per_kind ? LayerMode{modes[4 * l], modes[4 * l + 1], modes[4 * l + 2], modes[4 * l + 3]} : LayerMode{modes[l], modes[l], modes[l], modes[l]};
    if (!layer_mode_ok(lm[l]))
      return fail(c, ZK_E_ARG, "zk_model_set_layer_modes: layer %d: modes (%d, %d, %d, %d) — each must be ZK_F16 / ZK_F16C8 / ZK_F16X3, "
                  "ZK_F16C8 for QK^T needs the ZK_F16C8 QKV GEMM (it writes k's c8 plane), a split QK^T a split QKV GEMM",
                  l, lm[l].qkv, lm[l].att, lm[l].o, lm[l].mlp);
  }
  sm.mode = ZK_F16MIX;
  for (int l = 0; l < sm.n_layers; ++l) sm.layer_mode[l] = lm[l];
  return ZK_OK;
}

int zk_model_set_fx(zk_ctx* c, int stage, float fx_mean, float fx_std) {
  int rc = check_stage(c, stage);
  if (rc) return rc;
  if (!(fx_std > 0.f)) return fail(c, ZK_E_ARG, "fx_std must be > 0");
  c->model[stage].mean = fx_mean;
  c->model[stage].std = fx_std;
  return ZK_OK;
}

int zk_logmel(zk_ctx* c, const float* audio, int64_t n_samples, int64_t first_start, int64_t hop, int32_t win,
              int32_t n_windows) {
  if (!c) return ZK_E_ARG;
  if (!audio) {      // the audio slot (zk_audio_load)
    if (c->audio_slot_n <= 0) return fail(c, ZK_E_STATE, "audio is NULL and the audio slot is empty (zk_audio_load)");
    audio = c->audio_slot.as<float>();
    n_samples = c->audio_slot_n;
  }
  if (n_samples <= 0) return fail(c, ZK_E_ARG, "audio is empty");
  if (n_windows < 0 || hop < 0 || first_start < 0) return fail(c, ZK_E_ARG, "negative window geometry");
  const int nf = n_frames_for(win);
  if (nf <= 0) return fail(c, ZK_E_SHAPE, "window of %d samples is shorter than one 400-sample frame", win);
  HIPCHK(c, hipSetDevice(c->device));
  const void* d_audio = nullptr;
  int rc = to_device(c, audio, (size_t)n_samples * 4, c->audio_dev, &d_audio);
  if (rc) return rc;
  HIPCHK(c, c->feat.ensure((size_t)(n_windows > 0 ? n_windows : 1) * nf * ZK_NMEL * 4));
  c->feat_windows = n_windows;
  c->feat_frames = nf;
  {
    ProfScope ps(c, P_LOGMEL);
    zk_launch_logmel((const float*)d_audio, n_samples, first_start, hop, win, n_windows, nf, c->d_hann, c->d_tw, c->d_mel,
                     c->d_mel_lo, c->d_mel_hi, c->feat.as<float>(), c->stream);
  }
  HIPCHK(c, hipGetLastError());
  return finish(c);
}

int zk_features_expand(zk_ctx* c, float mean, float std, int32_t do_normalize, float* out) {
  if (!c || !out) return ZK_E_ARG;
  if (do_normalize && !(std > 0.f)) return fail(c, ZK_E_ARG, "std must be > 0");
  if (!do_normalize) { mean = 0.f; std = 1.f; }
  if (c->feat_windows <= 0) return ZK_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t bytes = (size_t)c->feat_windows * ZK_MAXLEN * ZK_NMEL * 4;
  float* d_out = out;
  const bool dev = is_device_ptr(out);
  if (!dev) { HIPCHK(c, c->st_out.ensure(bytes)); d_out = c->st_out.as<float>(); }
  zk_launch_expand_features(c->feat.as<float>(), c->feat_frames, c->feat_windows, mean, std * 2.0f, do_normalize, d_out,
                            c->stream);
  HIPCHK(c, hipGetLastError());
  if (!dev) return from_device(c, d_out, out, bytes);
  return finish(c);
}

int zk_features_get(zk_ctx* c, float* out, int32_t* n_windows, int32_t* n_frames) {
  if (!c) return ZK_E_ARG;
  if (n_windows) *n_windows = c->feat_windows;
  if (n_frames) *n_frames = c->feat_frames;
  if (out && c->feat_windows > 0)
    return from_device(c, c->feat.p, out, (size_t)c->feat_windows * c->feat_frames * ZK_NMEL * 4);
  return ZK_OK;
}

int zk_features_set(zk_ctx* c, const float* feats, int32_t n_windows, int32_t n_frames) {
  if (!c) return ZK_E_ARG;
  if (n_windows < 0 || n_frames < 1 || n_frames > ZK_MAXLEN) return fail(c, ZK_E_SHAPE, "feature slot of %d windows x %d frames (1..%d frames)", n_windows, n_frames, ZK_MAXLEN);
  if (n_windows > 0 && !feats) return fail(c, ZK_E_ARG, "feats is NULL");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t bytes = (size_t)n_windows * n_frames * ZK_NMEL * 4;
  HIPCHK(c, c->feat.ensure(bytes ? bytes : 4));
  if (bytes)
    HIPCHK(c, hipMemcpyAsync(c->feat.p, feats, bytes, is_device_ptr(feats) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
  c->feat_windows = n_windows;
  c->feat_frames = n_frames;
  // (a pageable host source is staged by the runtime before hipMemcpyAsync returns; pinned memory needs the sync)
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ZK_OK;
}

int zk_ast_forward(zk_ctx* c, int stage, const float* input_values, const int32_t* win_idx, int32_t B, float* logits) {
  int rc = check_stage(c, stage);
  if (rc) return rc;
  if (B < 0) return fail(c, ZK_E_ARG, "B < 0");
  if (B == 0) return ZK_OK;
  if (!logits) return fail(c, ZK_E_ARG, "logits is NULL");
  HIPCHK(c, hipSetDevice(c->device));
  StageModel& sm = c->model[stage];
  const void* d_in = nullptr;
  const void* d_idx = nullptr;
  if (input_values) {
    if ((rc = to_device(c, input_values, (size_t)B * ZK_MAXLEN * ZK_NMEL * 4, c->st_in, &d_in))) return rc;
  } else {
    if (c->feat_windows <= 0) return fail(c, ZK_E_STATE, "input_values is NULL and the feature slot is empty (zk_logmel)");
    if (!win_idx && B > c->feat_windows) return fail(c, ZK_E_SHAPE, "B=%d exceeds the %d windows in the feature slot", B, c->feat_windows);
    if (win_idx && (rc = to_device(c, win_idx, (size_t)B * 4, c->st_idx, &d_idx))) return rc;
  }
  const size_t lbytes = (size_t)B * sm.num_labels * 4;
  const bool dev_out = is_device_ptr(logits);
  float* d_logits = logits;
  if (!dev_out) { HIPCHK(c, c->st_out.ensure(lbytes)); d_logits = c->st_out.as<float>(); }
  if ((rc = forward_device(c, stage, (const float*)d_in, (const int32_t*)d_idx, B, d_logits))) return rc;
  if (!dev_out) return from_device(c, d_logits, logits, lbytes);
  return finish(c);
}

int zk_softmax(zk_ctx* c, const float* logits, int32_t n, int32_t num_labels, float* probs) {
  if (!c || !logits || !probs) return ZK_E_ARG;
  if (n <= 0) return ZK_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t bytes = (size_t)n * num_labels * 4;
  const void* d_in = nullptr;
  int rc = to_device(c, logits, bytes, c->st_in, &d_in);
  if (rc) return rc;
  const bool dev = is_device_ptr(probs);
  float* d_out = probs;
  if (!dev) { HIPCHK(c, c->st_out.ensure(bytes)); d_out = c->st_out.as<float>(); }
  zk_launch_softmax2((const float*)d_in, n, num_labels, d_out, c->stream);
  HIPCHK(c, hipGetLastError());
  if (!dev) return from_device(c, d_out, probs, bytes);
  return finish(c);
}

int zk_gate(zk_ctx* c, const float* logits, int32_t n, float thr1, float fwd_min_prob, float* probs, int32_t* idx,
            int32_t* cnt) {
  if (!c || !logits || !idx || !cnt) return ZK_E_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  if (n <= 0) {
    const int32_t z = 0;
    if (is_device_ptr(cnt)) HIPCHK(c, hipMemcpy(cnt, &z, 4, hipMemcpyHostToDevice));
    else *cnt = 0;
    return ZK_OK;
  }
  const void* d_in = nullptr;
  int rc = to_device(c, logits, (size_t)n * 8, c->st_in, &d_in);
  if (rc) return rc;
  HIPCHK(c, c->gate_idx.ensure((size_t)n * 4));
  HIPCHK(c, c->gate_cnt.ensure(16));
  float* d_probs = nullptr;
  const bool pdev = probs && is_device_ptr(probs);
  if (probs) { if (pdev) d_probs = probs; else { HIPCHK(c, c->st_out.ensure((size_t)n * 8)); d_probs = c->st_out.as<float>(); } }
  zk_launch_gate((const float*)d_in, n, thr1, fwd_min_prob, d_probs, c->gate_idx.as<int32_t>(), c->gate_cnt.as<int32_t>(), c->stream);
  HIPCHK(c, hipGetLastError());
  if (probs && !pdev && (rc = from_device(c, d_probs, probs, (size_t)n * 8))) return rc;
  if ((rc = from_device(c, c->gate_cnt.p, cnt, 4))) return rc;
  if ((rc = from_device(c, c->gate_idx.p, idx, (size_t)n * 4))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ZK_OK;
}

int zk_two_stage(zk_ctx* c, const float* audio, int64_t n_samples, int64_t first_start, int64_t hop, int32_t win,
                 int32_t N, float thr1, float fwd_min_prob, float* s1_logits, int32_t* swallow_idx, int32_t* n_swallow,
                 float* s2_logits) {
  int rc = check_stage(c, 0);
  if (!rc) rc = check_stage(c, 1);
  if (rc) return rc;
  if (!s1_logits || !swallow_idx || !n_swallow || !s2_logits) return fail(c, ZK_E_ARG, "NULL output");
  if (c->model[0].num_labels != 2 || c->model[1].num_labels != 2) return fail(c, ZK_E_SHAPE, "cascade needs 2-label heads");
  if (N <= 0) { int32_t z = 0; if (is_device_ptr(n_swallow)) HIPCHK(c, hipMemcpy(n_swallow, &z, 4, hipMemcpyHostToDevice)); else *n_swallow = 0; return ZK_OK; }
  const bool was_async = c->async;
  c->async = true;
  rc = zk_logmel(c, audio, n_samples, first_start, hop, win, N);
  c->async = was_async;
  if (rc) return rc;
  HIPCHK(c, c->s1_logits.ensure((size_t)N * 8));
  HIPCHK(c, c->s2_logits.ensure((size_t)N * 8));
  HIPCHK(c, c->gate_idx.ensure((size_t)N * 4));
  HIPCHK(c, c->gate_cnt.ensure(16));
  if ((rc = forward_device(c, 0, nullptr, nullptr, N, c->s1_logits.as<float>()))) return rc;
  zk_launch_gate(c->s1_logits.as<float>(), N, thr1, fwd_min_prob, nullptr, c->gate_idx.as<int32_t>(),
                 c->gate_cnt.as<int32_t>(), c->stream);
  HIPCHK(c, hipGetLastError());
  int32_t K = 0;
  HIPCHK(c, hipMemcpyAsync(&K, c->gate_cnt.p, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));  // the only host sync of the cascade: K sizes the stage-2 grid
  if (K > 0 && (rc = forward_device(c, 1, nullptr, c->gate_idx.as<int32_t>(), K, c->s2_logits.as<float>()))) return rc;
  if ((rc = from_device(c, c->s1_logits.p, s1_logits, (size_t)N * 8))) return rc;
  if ((rc = from_device(c, c->gate_idx.p, swallow_idx, (size_t)K * 4))) return rc;
  if (K > 0 && (rc = from_device(c, c->s2_logits.p, s2_logits, (size_t)K * 8))) return rc;
  if (is_device_ptr(n_swallow)) HIPCHK(c, hipMemcpyAsync(n_swallow, c->gate_cnt.p, 4, hipMemcpyDeviceToDevice, c->stream));
  else *n_swallow = K;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ZK_OK;
}

static int64_t resampled_len(int64_t n_in, int32_t orig_sr, int32_t new_sr) {
  int a = orig_sr, b = new_sr;
  while (b) { int t = a % b; a = b; b = t; }
  const int64_t orig = orig_sr / a, neu = new_sr / a;
  return (neu * n_in + orig - 1) / orig;
}

int zk_audio_load(zk_ctx* c, const void* data, int64_t n_bytes, int32_t format_tag, int32_t bits, int32_t channels,
                  int32_t sr, int32_t target_sr, int64_t* n_samples_out) {
  if (!c || !data) return ZK_E_ARG;
  if (channels < 1 || bits < 8 || sr <= 0 || target_sr <= 0) return fail(c, ZK_E_ARG, "bad audio format");
  const int64_t n_frames = n_bytes / ((int64_t)channels * (bits / 8));
  c->audio_slot_n = 0;
  if (n_samples_out) *n_samples_out = 0;
  if (n_frames <= 0) return ZK_OK;
  const bool was_async = c->async;
  c->async = true;      // one sync at the end: upload -> decode -> resample stay queued on the stream
  int rc;
  if (sr == target_sr) {
    rc = hipSuccess == c->audio_slot.ensure((size_t)n_frames * 4) ? ZK_OK : fail(c, ZK_E_NOMEM, "audio buffer");
    if (!rc) rc = zk_wav_decode(c, data, n_bytes, format_tag, bits, channels, c->audio_slot.as<float>());
    if (!rc) c->audio_slot_n = n_frames;
  } else {
    const int64_t n_out = resampled_len(n_frames, sr, target_sr);
    rc = hipSuccess == c->tmp_f32.ensure((size_t)n_frames * 4) && hipSuccess == c->audio_slot.ensure((size_t)n_out * 4)
             ? ZK_OK : fail(c, ZK_E_NOMEM, "audio buffers");
    if (!rc) rc = zk_wav_decode(c, data, n_bytes, format_tag, bits, channels, c->tmp_f32.as<float>());
    if (!rc) rc = zk_resample(c, c->tmp_f32.as<float>(), n_frames, sr, target_sr, c->audio_slot.as<float>(), n_out);
    if (!rc) c->audio_slot_n = n_out;
  }
  c->async = was_async;
  if (rc) return rc;
  if (n_samples_out) *n_samples_out = c->audio_slot_n;
  return finish(c);
}

int zk_audio_get(zk_ctx* c, float* out, int64_t* n_samples) {
  if (!c) return ZK_E_ARG;
  if (n_samples) *n_samples = c->audio_slot_n;
  if (out && c->audio_slot_n > 0) return from_device(c, c->audio_slot.p, out, (size_t)c->audio_slot_n * 4);
  return ZK_OK;
}

int zk_wav_decode(zk_ctx* c, const void* data, int64_t n_bytes, int32_t format_tag, int32_t bits, int32_t channels, float* out) {
  if (!c || !data || !out) return ZK_E_ARG;
  const bool ok = (format_tag == 1 && (bits == 8 || bits == 16 || bits == 24 || bits == 32)) ||
                  (format_tag == 3 && (bits == 32 || bits == 64));
  if (!ok) return fail(c, ZK_E_ARG, "unsupported WAVE sample format: tag %d, %d bits", format_tag, bits);
  if (channels < 1 || channels > 64 || n_bytes < 0) return fail(c, ZK_E_ARG, "bad channel count / size");
  const int64_t n_frames = n_bytes / ((int64_t)channels * (bits / 8));
  if (n_frames == 0) return ZK_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const void* d_in = nullptr;
  int rc = to_device(c, data, (size_t)n_frames * channels * (bits / 8), c->st_in, &d_in);
  if (rc) return rc;
  const bool dev = is_device_ptr(out);
  float* d_out = out;
  if (!dev) { HIPCHK(c, c->st_out.ensure((size_t)n_frames * 4)); d_out = c->st_out.as<float>(); }
  { ProfScope ps(c, P_WAVDEC); zk_launch_wav_decode((const unsigned char*)d_in, n_frames, format_tag, bits, channels, d_out, c->stream); }
  HIPCHK(c, hipGetLastError());
  if (!dev) return from_device(c, d_out, out, (size_t)n_frames * 4);
  return finish(c);
}

int zk_resample(zk_ctx* c, const float* in, int64_t n_in, int32_t orig_sr, int32_t new_sr, float* out, int64_t n_out) {
  if (!c || !in || !out) return ZK_E_ARG;
  if (orig_sr <= 0 || new_sr <= 0 || n_in <= 0) return fail(c, ZK_E_ARG, "bad resample arguments");
  HIPCHK(c, hipSetDevice(c->device));
  int a = orig_sr, b = new_sr;
  while (b) { int t = a % b; a = b; b = t; }
  const int orig = orig_sr / a, neu = new_sr / a;
  const int64_t want = (neu * n_in + orig - 1) / orig;
  if (n_out != want) return fail(c, ZK_E_SHAPE, "n_out must be ceil(new*n_in/orig) = %lld", (long long)want);
  if (orig != c->rs_orig || neu != c->rs_new) {
    // torchaudio.functional._get_sinc_resample_kernel, sinc_interp_hann, lowpass_filter_width=6, rolloff=0.99
    const double lpw = 6.0, rolloff = 0.99;
    const double base_freq = (double)(orig < neu ? orig : neu) * rolloff;
    const int width = (int)ceil(lpw * orig / base_freq);
    const int klen = 2 * width + orig;
    // the kernel stages the input stretch of 256 outputs in LDS (zk_launch_resample): every pair of standard audio rates
    // fits (44.1 -> 16 kHz: 5.4 KB); a ratio whose reduced period runs into the tens of thousands does not
    if (((int64_t)(255 / neu + 1) * orig + klen) * 4 > 60 * 1024)
      return fail(c, ZK_E_SHAPE, "resampling %d -> %d Hz (period %d : %d) is not supported", orig_sr, new_sr, orig, neu);
    std::vector<float> k((size_t)neu * klen);
    for (int p = 0; p < neu; ++p)
      for (int j = 0; j < klen; ++j) {
        double t = ((double)(-p) / neu + (double)(j - width) / orig) * base_freq;
        if (t < -lpw) t = -lpw;
        if (t > lpw) t = lpw;
        const double wdw = cos(t * M_PI / lpw / 2.0);
        const double tt = t * M_PI;
        const double sinc = tt == 0.0 ? 1.0 : sin(tt) / tt;
        k[(size_t)p * klen + j] = (float)(sinc * wdw * wdw * (base_freq / orig));
      }
    HIPCHK(c, c->rs_kern.ensure(k.size() * 4));
    HIPCHK(c, hipMemcpy(c->rs_kern.p, k.data(), k.size() * 4, hipMemcpyHostToDevice));
    c->rs_orig = orig; c->rs_new = neu; c->rs_width = width; c->rs_klen = klen;
  }
  const void* d_in = nullptr;
  int rc = to_device(c, in, (size_t)n_in * 4, c->st_in, &d_in);
  if (rc) return rc;
  const bool dev = is_device_ptr(out);
  float* d_out = out;
  if (!dev) { HIPCHK(c, c->st_out.ensure((size_t)n_out * 4)); d_out = c->st_out.as<float>(); }
  { ProfScope ps(c, P_RESAMPLE); zk_launch_resample((const float*)d_in, n_in, orig, neu, c->rs_width, c->rs_kern.as<float>(), c->rs_klen, d_out, n_out, c->stream); }
  HIPCHK(c, hipGetLastError());
  if (!dev) return from_device(c, d_out, out, (size_t)n_out * 4);
  return finish(c);
}

// ---- measurement ------------------------------------------------------------------------------------------------
int zk_prof_begin(zk_ctx* c) {
  if (!c) return ZK_E_ARG;
  c->prof = true;
  c->ev_used.clear();
  for (int i = 0; i < P_N; ++i) { c->prof_ms[i] = 0; c->prof_n[i] = 0; c->prof_flops[i] = 0; }
  return ZK_OK;
}
int zk_prof_end(zk_ctx* c) {
  if (!c) return ZK_E_ARG;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (auto& u : c->ev_used) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev_pool[u.second].first, c->ev_pool[u.second].second) == hipSuccess) {
      c->prof_ms[u.first] += ms;
      c->prof_n[u.first] += 1;
    }
  }
  c->ev_used.clear();
  c->prof = false;
  return ZK_OK;
}
int zk_prof_get(zk_ctx* c, const char* name, double* ms, int64_t* launches) {
  if (!c || !name) return ZK_E_ARG;
  for (int i = 0; i < P_N; ++i)
    if (!strcmp(name, kProfNames[i])) { if (ms) *ms = c->prof_ms[i]; if (launches) *launches = c->prof_n[i]; return ZK_OK; }
  return fail(c, ZK_E_ARG, "unknown profile class '%s'", name);
}

int zk_prof_get_flops(zk_ctx* c, const char* name, double* flops) {
  if (!c || !name || !flops) return ZK_E_ARG;
  for (int i = 0; i < P_N; ++i)
    if (!strcmp(name, kProfNames[i])) { *flops = c->prof_flops[i]; return ZK_OK; }
  return fail(c, ZK_E_ARG, "unknown profile class '%s'", name);
}
int zk_set_prune_last_layer(zk_ctx* c, int enable) { if (!c) return ZK_E_ARG; c->prune_last = enable != 0; return ZK_OK; }
int zk_set_layer0_reuse(zk_ctx* c, int enable) { if (!c) return ZK_E_ARG; c->l0_reuse = enable != 0; return ZK_OK; }

int zk_debug_set_tap(zk_ctx* c, int32_t layer) { if (!c) return ZK_E_ARG; c->tap_layer = layer; c->tap_windows = 0; return ZK_OK; }
int zk_debug_get_tap(zk_ctx* c, float* out, int32_t n_windows) {
  if (!c || !out) return ZK_E_ARG;
  if (n_windows > c->tap_windows) return fail(c, ZK_E_STATE, "tap holds %d windows, %d requested", c->tap_windows, n_windows);
  return from_device(c, c->tap.p, out, (size_t)n_windows * ZK_SEQ * ZK_HIDDEN * 4);
}

// ---- test hooks: run ONE kernel on caller-provided fp32 host data (tests/test_kernels_gpu.py) -------------------------
// ZK_TEST_TILED_IN / ZK_TEST_TILED_OUT (ORed into `epi` of zk_test_gemm / `nsplit` of zk_test_layernorm, zk_test_attention):
// the kernel reads its x planes / writes its output planes in the k-slice-major tile form (zk_planes::tiled) that the
// ZK_F16C8 forward uses between its GEMM-side kernels; the hook converts on the host, with the formula of zk_common.h
// written out independently, so that tests can demand tiled == row-major bit for bit.
}      // extern "C"
namespace {
size_t host_tiled_off(size_t m, size_t k, size_t K) {
  return (((m >> 8) * (K >> 6) + (k >> 6)) * 256 + (m & 255)) * 64 + ((((k & 63) >> 3) ^ ((m >> 1) & 7)) << 3) + (k & 7);
}
// device plane [rows_pad, K] (16-bit elements), row-major <-> tiled, through the host
int retile_plane(zk_ctx* c, half_t* d, size_t rows_pad, size_t K, bool to_tiled) {
  std::vector<uint16_t> a(rows_pad * K), b(rows_pad * K);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(a.data(), d, a.size() * 2, hipMemcpyDeviceToHost));
  for (size_t m = 0; m < rows_pad; ++m)
    for (size_t k = 0; k < K; ++k) {
      const size_t t = host_tiled_off(m, k, K), r = m * K + k;
      if (to_tiled) b[t] = a[r]; else b[r] = a[t];
    }
  HIPCHK(c, hipMemcpy(d, b.data(), b.size() * 2, hipMemcpyHostToDevice));
  return ZK_OK;
}
}      // namespace
extern "C" {

int zk_test_layernorm(zk_ctx* c, const float* x, const float* gamma, const float* beta, int32_t rows, float eps,
                      int32_t nsplit_flags, float* out) {
  if (!c) return ZK_E_ARG;
  const int32_t nsplit = nsplit_flags & 0xFF;
  const bool tiled_out = (nsplit_flags & ZK_TEST_TILED_OUT) != 0;
  if (tiled_out && nsplit != ZK_F16C8) return fail(c, ZK_E_ARG, "tiled planes exist in ZK_F16C8 only");
  HIPCHK(c, hipSetDevice(c->device));
  DevArena mem(c->stream);
  const size_t rows_pad = ((size_t)rows + 255) / 256 * 256;
  const size_t n = (size_t)rows * ZK_HIDDEN, np = rows_pad * ZK_HIDDEN;
  float *dx, *dg, *db; half_t *hi, *lo; int32_t* dexp;
  HIPCHK(c, mem.alloc(&dx, n * 4)); HIPCHK(c, mem.alloc(&dg, ZK_HIDDEN * 4)); HIPCHK(c, mem.alloc(&db, ZK_HIDDEN * 4));
  HIPCHK(c, mem.alloc(&hi, np * 2)); HIPCHK(c, mem.alloc(&lo, np * 2));
  HIPCHK(c, hipMemsetAsync(hi, 0, np * 2, c->stream)); HIPCHK(c, hipMemsetAsync(lo, 0, np * 2, c->stream));
  HIPCHK(c, mem.alloc(&dexp, (size_t)rows * 4)); HIPCHK(c, hipMemsetAsync(dexp, 0, (size_t)rows * 4, c->stream));
  HIPCHK(c, hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(dg, gamma, ZK_HIDDEN * 4, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(db, beta, ZK_HIDDEN * 4, hipMemcpyHostToDevice));
  const int lf = nsplit == ZK_F16C8 ? ZK_LO_C8 : ZK_LO_F16;
  {
    zk_planes op{hi, nsplit != ZK_F16 ? lo : nullptr, lf, dexp};
    op.tiled = tiled_out ? 1 : 0;
    zk_launch_layernorm(dx, ZK_HIDDEN, dg, db, rows, op, eps, c->stream);
  }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (tiled_out) {
    int rc = retile_plane(c, hi, rows_pad, ZK_HIDDEN, false);
    if (!rc) rc = retile_plane(c, lo, rows_pad, ZK_HIDDEN, false);
    if (rc) return rc;
  }
  std::vector<uint16_t> h(n), l(n, 0);
  std::vector<int32_t> ex(rows, 0);
  HIPCHK(c, hipMemcpy(h.data(), hi, n * 2, hipMemcpyDeviceToHost));
  if (nsplit != ZK_F16) HIPCHK(c, hipMemcpy(l.data(), lo, n * 2, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(ex.data(), dexp, (size_t)rows * 4, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (size_t i = 0; i < n; ++i) {      // planes hold y·2^-s of the row (s = 0 outside ZK_F16C8)
    const float scaled = half_bits_to_float(h[i]) + (nsplit != ZK_F16 ? lo_entry_to_float(l[i], lf) : 0.f);
    if (lf == ZK_LO_C8 && !c8_value_byte_ok(l[i], scaled)) ++bad;
    out[i] = ldexpf(scaled, ex[i / ZK_HIDDEN]);
  }
  if (lf == ZK_LO_C8)      // the row's largest plane entry must sit in (112, 224]
    for (int r = 0; r < rows; ++r) {
      float mx = 0.f;
      for (int k = 0; k < ZK_HIDDEN; ++k) mx = fmaxf(mx, fabsf(half_bits_to_float(h[(size_t)r * ZK_HIDDEN + k])));
      if (mx > 0.f && !(mx > 111.9f && mx <= 224.1f)) ++bad;
    }
  if (bad) return fail(c, ZK_E_STATE, "layernorm c8 plane: %zu value bytes are not the fp8 rounding of the (row-scaled) output / rows off (112, 224]", bad);
  return ZK_OK;
}

// x [M,K], w [N,K], bias [N] fp32 host.  epi STORE/GELU: out [M,N] = planes summed.  RESID: out [M,N] in/out.
// PATCH: M must be a multiple of 1212, pos [1214,N], out [(M/1212)*1214, N] (rows 0,1 of each window untouched).
int zk_test_gemm(zk_ctx* c, const float* x, const float* w, const float* bias, int32_t M, int32_t N, int32_t K,
                 int32_t epi_flags, int32_t nsplit, const float* pos, float* out) {
  if (!c) return ZK_E_ARG;
  const int32_t epi = epi_flags & 0xFF;
  const bool tiled_in = (epi_flags & ZK_TEST_TILED_IN) != 0, tiled_out = (epi_flags & ZK_TEST_TILED_OUT) != 0;
  const int pad_byte = (epi_flags & ZK_TEST_POISON_PAD) ? 0x7E : 0;      // 0x7E7E = fp16 NaN; as e4m3 bytes: 448
  if ((tiled_in || tiled_out) && nsplit != ZK_F16C8) return fail(c, ZK_E_ARG, "tiled planes exist in ZK_F16C8 only");
  if (tiled_out && epi != ZK_EPI_GELU) return fail(c, ZK_E_ARG, "only the GELU epilogue writes tiled planes");
  if (N % 256 || K % 64 || M < 1) return fail(c, ZK_E_SHAPE, "zk_test_gemm: need N%%256==0, K%%64==0");
  if (epi == ZK_EPI_PATCH && (M % ZK_NPATCH || N != ZK_HIDDEN)) return fail(c, ZK_E_SHAPE, "PATCH epilogue: M%%1212==0, N==768");
  HIPCHK(c, hipSetDevice(c->device));
  DevArena mem(c->stream);
  const size_t nx = (size_t)M * K, nw = (size_t)N * K;
  const size_t orows = epi == ZK_EPI_PATCH ? (size_t)(M / ZK_NPATCH) * ZK_SEQ : (size_t)M;
  const size_t no = orows * N;
  float *dx, *dw, *dbias, *dres = nullptr, *dpos = nullptr; half_t *xh, *xl, *wh, *wl, *oh = nullptr, *ol = nullptr;
  int32_t* dexp = nullptr;
  HIPCHK(c, mem.alloc(&dx, nx * 4)); HIPCHK(c, mem.alloc(&dw, nw * 4)); HIPCHK(c, mem.alloc(&dbias, (size_t)N * 4));
  // x planes are padded by one 256-row tile: the ZK_F16C8 kernel reads the last row block whole (zk_gemm_args: rows
  // M .. ceil(M/256)*256 must be readable; their products are never stored)
  const size_t m_pad = ((size_t)M + 255) / 256 * 256;
  const size_t nxp = m_pad * K;      // exactly the whole row blocks the launch contract asks for (zk_gemm_args::x_rows)
  HIPCHK(c, mem.alloc(&xh, nxp * 2)); HIPCHK(c, mem.alloc(&xl, nxp * 2));
  // (memsets go on the context's stream: a plain hipMemset runs on the NULL stream, which the non-blocking context stream
  // does not wait for — it could land after the kernels below had written the buffer)
  HIPCHK(c, hipMemsetAsync(xh, pad_byte, nxp * 2, c->stream)); HIPCHK(c, hipMemsetAsync(xl, pad_byte, nxp * 2, c->stream));
  HIPCHK(c, mem.alloc(&wh, nw * 2)); HIPCHK(c, mem.alloc(&wl, nw * 2));
  HIPCHK(c, hipMemcpy(dx, x, nx * 4, hipMemcpyHostToDevice)); HIPCHK(c, hipMemcpy(dw, w, nw * 4, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(dbias, bias, (size_t)N * 4, hipMemcpyHostToDevice));
  zk_launch_split_f32(dx, (int64_t)nx, 1.f, xh, xl, c->stream);
  zk_launch_split_f32(dw, (int64_t)nw, 1.f, wh, wl, c->stream);
  int w_exp = 0;
  if (nsplit == ZK_F16C8) {      // lo planes become c8 byte pairs
    w_exp = c8_exponent(w, nw);
    HIPCHK(c, mem.alloc(&dexp, (size_t)(M + 256) * 4)); HIPCHK(c, hipMemsetAsync(dexp, 0, (size_t)(M + 256) * 4, c->stream));
    zk_launch_split_rows_c8(dx, M, K, xh, xl, dexp, c->stream);      // row-scaled planes, as LayerNorm writes them
    zk_launch_split_c8(dw, (int64_t)nw, w_exp, 1, wl, c->stream);
  }
  if (epi == ZK_EPI_RESID || epi == ZK_EPI_PATCH) {
    HIPCHK(c, mem.alloc(&dres, no * 4));
    HIPCHK(c, hipMemcpy(dres, out, no * 4, hipMemcpyHostToDevice));
    if (epi == ZK_EPI_PATCH) { HIPCHK(c, mem.alloc(&dpos, (size_t)ZK_SEQ * N * 4)); HIPCHK(c, hipMemcpy(dpos, pos, (size_t)ZK_SEQ * N * 4, hipMemcpyHostToDevice)); }
  } else {
    const size_t nop = tiled_out ? m_pad * N : no;      // tiled planes hold whole row blocks
    HIPCHK(c, mem.alloc(&oh, nop * 2)); HIPCHK(c, mem.alloc(&ol, nop * 2));
    HIPCHK(c, hipMemsetAsync(oh, 0, nop * 2, c->stream)); HIPCHK(c, hipMemsetAsync(ol, 0, nop * 2, c->stream));
  }
  if (tiled_in) {
    int rc = retile_plane(c, xh, m_pad, K, true);
    if (!rc) rc = retile_plane(c, xl, m_pad, K, true);
    if (rc) return rc;
  }
  zk_gemm_args a;
  a.x_hi = xh; a.x_lo = nsplit != ZK_F16 ? xl : nullptr; a.w_hi = wh; a.w_lo = nsplit != ZK_F16 ? wl : nullptr; a.bias = dbias;
  a.x_rowexp = dexp;
  a.M = M; a.N = N; a.K = K; a.o_hi = oh; a.o_lo = nsplit != ZK_F16 ? ol : nullptr; a.resid = dres; a.pos = dpos; a.lo_n_limit = N; a.lo_c8_from = 1 << 30;
  a.w_exp = w_exp;
  // ZK_TEST_SHORT_X: state the allocation as M rows only — the launcher must then refuse an M that is not a multiple of 256
  a.x_rows = (epi_flags & ZK_TEST_SHORT_X) ? (int64_t)M : (int64_t)m_pad; a.x_tiled = tiled_in; a.o_tiled = tiled_out;
  if (nsplit == ZK_F16C8) {
    if (zk_launch_gemm_c8(a, epi, c->stream)) return fail(c, ZK_E_SHAPE, "zk_launch_gemm_c8 refused the launch");
  } else zk_launch_gemm(a, epi, nsplit, c->stream);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (tiled_out) {
    int rc = retile_plane(c, oh, m_pad, N, false);
    if (!rc) rc = retile_plane(c, ol, m_pad, N, false);
    if (rc) return rc;
  }
  if (dres) HIPCHK(c, hipMemcpy(out, dres, no * 4, hipMemcpyDeviceToHost));
  else {
    std::vector<uint16_t> h(no), l(no);
    HIPCHK(c, hipMemcpy(h.data(), oh, no * 2, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(l.data(), ol, no * 2, hipMemcpyDeviceToHost));
    const int lf = (nsplit == ZK_F16C8 && epi == ZK_EPI_GELU) ? ZK_LO_C8 : ZK_LO_F16;
    size_t bad = 0;
    for (size_t i = 0; i < no; ++i) {
      out[i] = half_bits_to_float(h[i]) + lo_entry_to_float(l[i], lf);
      if (lf == ZK_LO_C8 && !c8_value_byte_ok(l[i], out[i])) ++bad;
    }
    if (bad) return fail(c, ZK_E_STATE, "gemm GELU c8 plane: %zu value bytes are not the fp8 rounding of the output", bad);
  }
  return ZK_OK;
}

// qkv fp32 host [W*1214, 2304] -> out fp32 host [W*1214, 768]
int zk_test_attention(zk_ctx* c, const float* qkv, int32_t W, int32_t nsplit_flags, float* out) {
  if (!c) return ZK_E_ARG;
  const int32_t nsplit = nsplit_flags & 0xFF;
  const bool tiled_out = (nsplit_flags & ZK_TEST_TILED_OUT) != 0;
  if (tiled_out && nsplit != ZK_F16C8) return fail(c, ZK_E_ARG, "tiled planes exist in ZK_F16C8 only");
  HIPCHK(c, hipSetDevice(c->device));
  DevArena mem(c->stream);
  const size_t rows = (size_t)W * ZK_SEQ, nq = rows * 3 * ZK_HIDDEN, no = rows * ZK_HIDDEN;
  const size_t rows_pad = (rows + 255) / 256 * 256, nop = rows_pad * ZK_HIDDEN;
  float* dq; half_t *qh, *ql, *oh, *ol;
  HIPCHK(c, mem.alloc(&dq, nq * 4)); HIPCHK(c, mem.alloc(&qh, nq * 2)); HIPCHK(c, mem.alloc(&ql, nq * 2));
  HIPCHK(c, mem.alloc(&oh, nop * 2)); HIPCHK(c, mem.alloc(&ol, nop * 2));
  HIPCHK(c, hipMemsetAsync(oh, 0, nop * 2, c->stream)); HIPCHK(c, hipMemsetAsync(ol, 0, nop * 2, c->stream));
  HIPCHK(c, hipMemcpy(dq, qkv, nq * 4, hipMemcpyHostToDevice));
  zk_launch_split_f32(dq, (int64_t)nq, 1.f, qh, ql, c->stream);
  const int lf = nsplit == ZK_F16C8 ? ZK_LO_C8 : ZK_LO_F16;
  if (nsplit == ZK_F16C8)      // as the fused QKV epilogue of this mode leaves them: k's lo entries are c8 byte pairs
    zk_launch_split_c8_cols(dq, (int)rows, 3 * ZK_HIDDEN, ZK_HIDDEN, ZK_HIDDEN, ql, c->stream);
  {
    zk_planes op{oh, nsplit != ZK_F16 ? ol : nullptr, lf};
    op.tiled = tiled_out ? 1 : 0;
    zk_launch_attention(zk_planes{qh, nsplit != ZK_F16 ? ql : nullptr, ZK_LO_F16}, op, W,
                        nsplit == ZK_F16C8 ? 2 : (nsplit != ZK_F16 ? 3 : 1), 0, c->stream);
  }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (tiled_out) {
    int rc = retile_plane(c, oh, rows_pad, ZK_HIDDEN, false);
    if (!rc) rc = retile_plane(c, ol, rows_pad, ZK_HIDDEN, false);
    if (rc) return rc;
  }
  std::vector<uint16_t> h(no), l(no);
  HIPCHK(c, hipMemcpy(h.data(), oh, no * 2, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(l.data(), ol, no * 2, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (size_t i = 0; i < no; ++i) {
    out[i] = half_bits_to_float(h[i]) + lo_entry_to_float(l[i], lf);
    if (lf == ZK_LO_C8 && !c8_value_byte_ok(l[i], out[i])) ++bad;
  }
  if (bad) return fail(c, ZK_E_STATE, "attention c8 plane: %zu value bytes are not the fp8 rounding of the output", bad);
  return ZK_OK;
}

int zk_test_split_c8(zk_ctx* c, const float* x, int64_t n, int32_t w_exp, int32_t is_weight, uint16_t* out) {
  if (!c || !x || !out || n <= 0) return ZK_E_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  DevArena mem(c->stream);
  float* dx; half_t* dc;
  HIPCHK(c, mem.alloc(&dx, (size_t)n * 4)); HIPCHK(c, mem.alloc(&dc, (size_t)n * 2));
  HIPCHK(c, hipMemcpy(dx, x, (size_t)n * 4, hipMemcpyHostToDevice));
  zk_launch_split_c8(dx, n, w_exp, is_weight, dc, c->stream);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(out, dc, (size_t)n * 2, hipMemcpyDeviceToHost));
  return ZK_OK;
}

}  // extern "C"
