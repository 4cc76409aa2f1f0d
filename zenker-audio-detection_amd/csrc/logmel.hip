// Kaldi-style log-mel filter bank of ASTFeatureExtractor, numpy branch, straight from the long recording.
// Replaces  window_audio (src/test_long_audio_windows_2stage.py:62-75)  +  _extract_fbank_features
// ($TF/.../feature_extraction_audio_spectrogram_transformer.py:124-141)  +  spectrogram ($TF/audio_utils.py:956-1015):
//   frame (400 @ hop 160, center=False) -> fp64: remove DC, pre-emphasis 0.97, symmetric Hann-400, zero-pad 512,
//   rFFT -> complex64 storage -> |.|^2 (fp64) -> mel (257x128, fp64) -> max(1.19e-7) -> log -> fp32.
// The reference does all of this in float64, and so does this kernel: fp64 costs nothing here (7.6 MFLOP per window
// against 261 GFLOP for the transformer) and keeps parity at the last-ulp level instead of fp32-FFT noise on weak bins.
//
// One 256-thread workgroup per frame; windows are never materialised: frame f of window w reads
// audio[first_start + w*hop + f*160 ...] (coalesced; neighbouring frames/windows overlap, so re-reads hit L2).
// The 512-point FFT runs radix-2 in LDS (one butterfly per thread per stage).
// Output is the COMPACT un-normalised feature [n_windows, n_frames, 128]; padding to 1024 rows and the
// (x-mean)/(2 std) normalisation are applied by the consumers (embed.hip / zk_expand_features).
#include "zk_common.h"

namespace {

__global__ __launch_bounds__(256) void logmel_kernel(const float* __restrict__ audio, int64_t n_samples,
                                                     int64_t first_start, int64_t hop, int n_frames,
                                                     const double* __restrict__ hann,
                                                     const double* __restrict__ twiddle,  // [256][2] cos,-sin
                                                     const double* __restrict__ mel,      // [257][128]
                                                     const int32_t* __restrict__ mel_lo,
                                                     const int32_t* __restrict__ mel_hi, float* __restrict__ out) {
  __shared__ double re[ZK_FFT];
  __shared__ double im[ZK_FFT];
  __shared__ double red[256];
  const int tid = threadIdx.x;
  const int w = blockIdx.x / n_frames;
  const int f = blockIdx.x - w * n_frames;
  const int64_t s0 = first_start + (int64_t)w * hop + (int64_t)f * ZK_FRAME_HOP;

  // ---- load 400 samples (zero beyond the recording: window_audio zero-pads a short recording) ----
  double x0 = 0.0, x1 = 0.0;  // samples tid and tid+256
  {
    const int64_t i0 = s0 + tid;
    if (i0 < n_samples) x0 = (double)audio[i0];
    if (tid + 256 < ZK_FRAME_LEN) {
      const int64_t i1 = s0 + tid + 256;
      if (i1 < n_samples) x1 = (double)audio[i1];
    }
  }
  red[tid] = x0 + x1;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const double mean = red[0] / (double)ZK_FRAME_LEN;
  // mean-removed samples into re[] (natural order) so that the pre-emphasis can read its left neighbour
  re[tid] = x0 - mean;
  if (tid + 256 < ZK_FRAME_LEN) re[tid + 256] = x1 - mean;
  __syncthreads();
  double y0, y1 = 0.0;
  {
    const double c = re[tid];
    y0 = (tid == 0) ? c * (1.0 - 0.97) : c - 0.97 * re[tid - 1];
    y0 *= hann[tid];
    if (tid + 256 < ZK_FRAME_LEN) {
      y1 = (re[tid + 256] - 0.97 * re[tid + 255]) * hann[tid + 256];
    }
  }
  __syncthreads();
  // bit-reversed scatter (9 bits), zero imaginary part, zero padding 400..511
  re[__brev((unsigned)tid) >> 23] = y0;
  re[__brev((unsigned)(tid + 256)) >> 23] = y1;
  im[tid] = 0.0;
  im[tid + 256] = 0.0;
  __syncthreads();

  // ---- 512-point radix-2 DIT FFT ----
#pragma unroll
  for (int st = 0; st < 9; ++st) {
    const int hlf = 1 << st;
    const int pos = tid & (hlf - 1);
    const int i = ((tid >> st) << (st + 1)) + pos;
    const int j = i + hlf;
    const int tw = pos << (8 - st);
    const double wr = twiddle[2 * tw], wi = twiddle[2 * tw + 1];
    const double ar = re[i], ai = im[i];
    const double br = re[j], bi = im[j];
    const double tr = br * wr - bi * wi;
    const double ti = br * wi + bi * wr;
    re[i] = ar + tr; im[i] = ai + ti;
    re[j] = ar - tr; im[j] = ai - ti;
    __syncthreads();
  }

  // ---- power spectrum of the complex64-rounded bins, in place into red[] / re[256] ----
  double p0, p256 = 0.0;
  {
    const float r32 = (float)re[tid], i32 = (float)im[tid];
    p0 = (double)r32 * (double)r32 + (double)i32 * (double)i32;
    if (tid == 0) {
      const float rr = (float)re[256], ii = (float)im[256];
      p256 = (double)rr * (double)rr + (double)ii * (double)ii;
    }
  }
  __syncthreads();
  re[tid] = p0;
  if (tid == 0) re[256] = p256;
  __syncthreads();

  // ---- mel filter bank + floor + log ----
  if (tid < ZK_NMEL) {
    double acc = 0.0;
    const int lo = mel_lo[tid], hi = mel_hi[tid];
    for (int k = lo; k < hi; ++k) acc = fma(re[k], mel[(size_t)k * ZK_NMEL + tid], acc);
    acc = fmax(acc, 1.192092955078125e-07);
    out[((size_t)w * n_frames + f) * ZK_NMEL + tid] = (float)log(acc);
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
  hipLaunchKernelGGL(logmel_kernel, dim3((unsigned)(n_windows * n_frames)), dim3(256), 0, s, audio, n_samples,
                     first_start, hop, n_frames, hann, twiddle, mel, mel_lo, mel_hi, out);
}

void zk_launch_expand_features(const float* feats, int n_frames, int n_windows, float mean, float std2,
                               int do_normalize, float* out, hipStream_t s) {
  if (n_windows <= 0) return;
  const int64_t total = (int64_t)n_windows * ZK_MAXLEN * (ZK_NMEL / 4);
  hipLaunchKernelGGL(expand_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, feats, n_frames,
                     n_windows, mean, std2, do_normalize, out);
}
