// Kaldi-style log-mel filter bank of ASTFeatureExtractor, numpy branch, straight from the long recording.
// Replaces  window_audio (src/test_long_audio_windows_2stage.py:62-75)  +  _extract_fbank_features
// ($TF/.../feature_extraction_audio_spectrogram_transformer.py:124-141)  +  spectrogram ($TF/audio_utils.py:956-1015):
//   frame (400 @ hop 160, center=False) -> fp64: remove DC, pre-emphasis 0.97, symmetric Hann-400, zero-pad 512,
//   rFFT -> complex64 storage -> |.|^2 (fp64) -> mel (257x128, fp64) -> max(1.19e-7) -> log -> fp32.
// The reference does all of this in float64, and so does this kernel: fp64 costs nothing here (7.6 MFLOP per window
// against 261 GFLOP for the transformer) and keeps parity at the last-ulp level instead of fp32-FFT noise on weak bins.
//
// gfx950 structure (round 5): PERSISTENT 256-thread workgroups, ONE WAVE PER FRAME.  A workgroup stages the tables every
// frame needs into LDS once — the 256 FFT twiddles, the Hann window, and the mel filter bank as its BAND (a triangle covers
// a few to a few dozen of the 257 bins: some 500 non-zero weights instead of 257 x 128) — and then walks groups of four consecutive frames;
// wave v of the workgroup owns frame 4g + v from its first sample to its 128 log-mel values.  Nothing a frame computes is
// shared with another wave, so the only synchronisation inside the frame loop is wave-local (LDS is processed in order
// per wave: a fence for the compiler, no s_barrier).  The 512-point radix-2 FFT keeps THREE stages at a time in registers
// (eight values per lane, lm_group8): stages 0-2 on the samples the lane loaded itself, two LDS exchanges in all instead of
// nine read-modify-write sweeps — the old kernel (one workgroup per frame, one butterfly per thread and stage, 24 block
// barriers) moved ~200 KB through LDS per frame at stride-2^k bank conflicts and took 0.54 ms per 1 024 windows (2.7 % of the
// HBM peak on its algorithmic bytes); taking the barriers out alone changed nothing (0.59 ms): the LDS sweeps were the time.
// Windows are never materialised: frame f of window w reads audio[first_start + w*hop + f*160 ...] (coalesced; neighbouring
// frames / windows overlap, so re-reads hit L2).  Output is the COMPACT un-normalised feature [n_windows, n_frames, 128];
// padding to 1024 rows and the (x-mean)/(2 std) normalisation are applied by the consumers (embed.hip / zk_expand_features).
#include "zk_common.h"

namespace {

constexpr int LM_FPW = 4;                 // frames per workgroup pass = waves per workgroup
constexpr int LM_BAND_MAX = ZK_MEL_BAND_MAX;         // >= the band's entries (the sum over filters of mel_hi - mel_lo: about 500 for the AST filter bank)

// wave-local ordering of LDS traffic: all 64 lanes run in lockstep and the LDS unit serves a wave's instructions in order, so
// a read issued after a write of the same wave sees it — the compiler only has to keep the order
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double wave_sum(double v) {      // all lanes get the sum (fixed butterfly order)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// position p of the FFT work array -> LDS slot: 8 doubles of padding per 64 keep the stride-8 (pass 2) and stride-64 (pass 3)
// gathers of a wave on 64 distinct 8-byte slots
__device__ __forceinline__ int lm_slot(int p) { return p + ((p >> 6) << 3); }
constexpr int LM_ARR = ZK_FFT + (ZK_FFT >> 6) * 8;      // 576

// Three consecutive radix-2 DIT stages on eight values held by one lane: local pairs (b, b+1), (b, b+2), (b, b+4); the
// twiddle index of stage s and local element b comes from tw(s, b).  The butterflies are those of the plain nine-stage
// radix-2 FFT (same operands, same table entries, same order per element): the grouping only keeps three stages in registers.
template <class TW>
__device__ __forceinline__ void lm_group8(double (&r)[8], double (&im)[8], const double* __restrict__ s_tw, TW tw) {
  auto bfly = [&](int x, int y, int t) __attribute__((always_inline)) {
    const double wr = s_tw[2 * t], wi = s_tw[2 * t + 1];
    const double ar = r[x], ai = im[x];
    const double br = r[y], bi = im[y];
    const double tr = br * wr - bi * wi;
    const double ti = br * wi + bi * wr;
    r[x] = ar + tr; im[x] = ai + ti;
    r[y] = ar - tr; im[y] = ai - ti;
  };
#pragma unroll
  for (int b = 0; b < 8; b += 2) bfly(b, b + 1, tw(0, b));
#pragma unroll
  for (int b = 0; b < 8; ++b) if (!(b & 2)) bfly(b, b + 2, tw(1, b));
#pragma unroll
  for (int b = 0; b < 4; ++b) bfly(b, b + 4, tw(2, b));
}

__global__ __launch_bounds__(256) void logmel_kernel(const float* __restrict__ audio, int64_t n_samples,
                                                     int64_t first_start, int64_t hop, int n_frames, int n_windows,
                                                     const double* __restrict__ hann,
                                                     const double* __restrict__ twiddle,  // [256][2] cos,-sin
                                                     const double* __restrict__ mel,      // [257][128]
                                                     const int32_t* __restrict__ mel_lo,
                                                     const int32_t* __restrict__ mel_hi, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) double re[LM_FPW][LM_ARR];
  __shared__ __attribute__((aligned(16))) double im[LM_FPW][LM_ARR];
  __shared__ double s_tw[ZK_FFT];            // [256][2]
  __shared__ double s_hann[ZK_FRAME_LEN];
  __shared__ double s_band[LM_BAND_MAX];     // filter m: weights of bins mel_lo[m] .. mel_hi[m]-1 at s_off[m] ..
  __shared__ int s_lo[ZK_NMEL], s_hi[ZK_NMEL], s_off[ZK_NMEL];
  const int tid = threadIdx.x, lane = tid & 63, v = tid >> 6;

  // ---- tables -> LDS, once per workgroup ----
  for (int i = tid; i < ZK_FFT; i += 256) s_tw[i] = twiddle[i];
  for (int i = tid; i < ZK_FRAME_LEN; i += 256) s_hann[i] = hann[i];
  if (tid == 0) {
    int off = 0;
    for (int m = 0; m < ZK_NMEL; ++m) { s_lo[m] = mel_lo[m]; s_hi[m] = mel_hi[m]; s_off[m] = off; off += mel_hi[m] - mel_lo[m]; }
  }
  __syncthreads();
  if (tid < ZK_NMEL)
    for (int k = s_lo[tid]; k < s_hi[tid]; ++k) s_band[s_off[tid] + k - s_lo[tid]] = mel[(size_t)k * ZK_NMEL + tid];
  __syncthreads();

  double* xr = re[v];
  double* xi = im[v];
  const int blk = (int)(__brev((unsigned)lane) >> 26);      // pass 1: this lane's block of eight positions (6-bit reversal)
  const int groups_per_window = (n_frames + LM_FPW - 1) / LM_FPW;
  const int64_t n_groups = (int64_t)n_windows * groups_per_window;
  for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const int w = (int)(g / groups_per_window);
    const int f = (int)(g - (int64_t)w * groups_per_window) * LM_FPW + v;
    if (f >= n_frames) continue;      // (wave-uniform; no block barrier below)
    const int64_t s0 = first_start + (int64_t)w * hop + (int64_t)f * ZK_FRAME_HOP;

    // ---- 400 samples (zero beyond the recording: window_audio zero-pads a short recording); lane holds samples lane + 64 j ----
    double x[7];
    double part = 0.0;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int i = lane + 64 * j;
      const int64_t gi = s0 + i;
      x[j] = (i < ZK_FRAME_LEN && gi < n_samples) ? (double)audio[gi] : 0.0;
      part += x[j];
    }
    const double mean = wave_sum(part) / (double)ZK_FRAME_LEN;
    // DC removal, pre-emphasis against the left neighbour (sample i-1 sits in lane-1, or in lane 63 of the previous j), Hann
    double r8[8], i8[8];      // pass-1 operands: block element b = 3-bit reversal of j
    {
      double y[8];
      double prev63 = 0.0;      // sample 64 j - 1 (mean removed)
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int i = lane + 64 * j;
        const double c = x[j] - mean;
        double left = __shfl_up(c, 1, 64);
        if (lane == 0) left = prev63;
        prev63 = __shfl(c, 63, 64);
        y[j] = 0.0;
        if (i < ZK_FRAME_LEN) y[j] = ((i == 0) ? c * (1.0 - 0.97) : c - 0.97 * left) * s_hann[i];
      }
      y[7] = 0.0;      // zero padding 400..511 (also the tail of j = 6)
      constexpr int rev3[8] = {0, 4, 2, 6, 1, 5, 3, 7};
#pragma unroll
      for (int b2 = 0; b2 < 8; ++b2) { r8[b2] = y[rev3[b2]]; i8[b2] = 0.0; }
    }
    // ---- 512-point radix-2 DIT FFT, three stages per pass in registers ----
    // pass 1 (stages 0-2): positions 8 blk + b are the samples (lane + 64 j) this lane already holds
    lm_group8(r8, i8, s_tw, [](int s, int b2) { return s == 0 ? 0 : (s == 1 ? (b2 & 1) << 7 : (b2 & 3) << 6); });
    {
      const int base = lm_slot(8 * blk);
#pragma unroll
      for (int b2 = 0; b2 < 8; ++b2) { xr[base + b2] = r8[b2]; xi[base + b2] = i8[b2]; }
    }
    wave_sync();
    // pass 2 (stages 3-5): positions 64 c + a + 8 b, lane = (a = lane & 7, c = lane >> 3)
    {
      const int a2 = lane & 7, c2 = lane >> 3;
#pragma unroll
      for (int b2 = 0; b2 < 8; ++b2) { const int sl = lm_slot(64 * c2 + a2 + 8 * b2); r8[b2] = xr[sl]; i8[b2] = xi[sl]; }
      lm_group8(r8, i8, s_tw, [a2](int s, int b2) {
        return s == 0 ? a2 << 5 : (s == 1 ? (a2 + 8 * (b2 & 1)) << 4 : (a2 + 8 * (b2 & 3)) << 3);
      });
#pragma unroll
      for (int b2 = 0; b2 < 8; ++b2) { const int sl = lm_slot(64 * c2 + a2 + 8 * b2); xr[sl] = r8[b2]; xi[sl] = i8[b2]; }
    }
    wave_sync();
    // pass 3 (stages 6-8): positions lane + 64 b; the results are the bins lane + 64 b
#pragma unroll
    for (int b2 = 0; b2 < 8; ++b2) { const int sl = lm_slot(lane + 64 * b2); r8[b2] = xr[sl]; i8[b2] = xi[sl]; }
    lm_group8(r8, i8, s_tw, [lane](int s, int b2) {
      return s == 0 ? lane << 2 : (s == 1 ? (lane + 64 * (b2 & 1)) << 1 : lane + 64 * (b2 & 3));
    });
    wave_sync();      // (every lane has read its pass-3 operands: xi[] may be overwritten)

    // ---- power spectrum of the complex64-rounded bins 0..256 -> xi[k] (natural index) ----
#pragma unroll
    for (int b2 = 0; b2 < 4; ++b2) {
      const float r32 = (float)r8[b2], i32 = (float)i8[b2];
      xi[lane + 64 * b2] = (double)r32 * (double)r32 + (double)i32 * (double)i32;
    }
    if (lane == 0) {
      const float rr = (float)r8[4], ii = (float)i8[4];
      xi[256] = (double)rr * (double)rr + (double)ii * (double)ii;
    }
    wave_sync();

    // ---- mel filter bank (band form, bins ascending as before) + floor + log: two filters per lane ----
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int m = lane + 64 * h;
      const int lo = s_lo[m], hi = s_hi[m];
      const double* wgt = s_band + s_off[m] - lo;
      double acc = 0.0;
      for (int k = lo; k < hi; ++k) acc = fma(xi[k], wgt[k], acc);
      acc = fmax(acc, 1.192092955078125e-07);
      out[((size_t)w * n_frames + f) * ZK_NMEL + m] = (float)log(acc);
    }
    wave_sync();      // (the next frame of this wave overwrites xr / xi)
  }
}

// compact [N, n_frames, 128] -> HF input_values [N, 1024, 128] (zero rows past n_frames, then normalise)
__global__ __launch_bounds__(256) void expand_kernel(const float* __restrict__ feats, int n_frames, int n_windows,
                                                     float mean, float std2, int do_normalize,
                                                     float* __restrict__ out) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one float4 each
  const int64_t total = (int64_t)n_windows * ZK_MAXLEN * (ZK_NMEL / 4);
  if (gid >= total) return;
  const int c4 = (int)(gid % (ZK_NMEL / 4));
  const int64_t r = gid / (ZK_NMEL / 4);
  const int row = (int)(r % ZK_MAXLEN);
  const int64_t w = r / ZK_MAXLEN;
  f4_t v = {0.f, 0.f, 0.f, 0.f};
  if (row < n_frames) v = *(const f4_t*)(feats + ((size_t)w * n_frames + row) * ZK_NMEL + c4 * 4);
  if (do_normalize) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (v[j] - mean) / std2;
  }
  *(f4_t*)(out + (size_t)gid * 4) = v;
}

}  // namespace

void zk_launch_logmel(const float* audio, int64_t n_samples, int64_t first_start, int64_t hop, int32_t win,
                      int n_windows, int n_frames, const double* hann, const double* twiddle, const double* mel,
                      const int32_t* mel_lo, const int32_t* mel_hi, float* out, hipStream_t s) {
  (void)win;
  if (n_windows <= 0 || n_frames <= 0) return;
  // persistent workgroups: up to 8 per CU (53 KiB of LDS each would allow 3; the register / wave limits 8 — 3 x 256 CUs)
  const int64_t n_groups = (int64_t)n_windows * ((n_frames + LM_FPW - 1) / LM_FPW);
  const int64_t cap = 256 * 3;
  hipLaunchKernelGGL(logmel_kernel, dim3((unsigned)(n_groups < cap ? n_groups : cap)), dim3(256), 0, s, audio, n_samples,
                     first_start, hop, n_frames, n_windows, hann, twiddle, mel, mel_lo, mel_hi, out);
}

void zk_launch_expand_features(const float* feats, int n_frames, int n_windows, float mean, float std2,
                               int do_normalize, float* out, hipStream_t s) {
  if (n_windows <= 0) return;
  const int64_t total = (int64_t)n_windows * ZK_MAXLEN * (ZK_NMEL / 4);
  hipLaunchKernelGGL(expand_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, feats, n_frames,
                     n_windows, mean, std2, do_normalize, out);
}
