// Small HBM-bound helpers: fp32 -> (hi, lo) fp16 plane split (weights at load time, test hooks) and the
// polyphase sinc resampler of load_audio (src/test_long_audio_windows_2stage.py:57-58).
#include "zk_common.h"

namespace {

__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ src, int64_t n, float scale,
                                                    half_t* __restrict__ hi, half_t* __restrict__ lo) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  if (i + 3 < n) {
    const f4_t v = *(const f4_t*)(src + i);
    h4_t h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x = v[j] * scale;
      zk_pin(x);
      h[j] = (half_t)x;
      l[j] = (half_t)(x - (float)h[j]);
    }
    *(h4_t*)(hi + i) = h;
    if (lo) *(h4_t*)(lo + i) = l;
  } else {
    for (int64_t k = i; k < n; ++k) {
      const float x = src[k] * scale;
      const half_t h = (half_t)x;
      hi[k] = h;
      if (lo) lo[k] = (half_t)(x - (float)h);
    }
  }
}

// fp32 -> c8 plane (16 bits per element, see zk_common.h).  Weights: (fp8(w·2^e), fp8((w - fp16(w))·2^(e+11)));
// activations: (fp8((x - fp16(x))·2^11), fp8(x)).  Element i of the c8 plane meets element i of the other operand's
// c8 plane in the fp8 MFMA, so byte 0 of one side always multiplies byte 0 of the other: w8·xl8 + wl8·x8.
__global__ __launch_bounds__(256) void split_c8_kernel(const float* __restrict__ src, int64_t n, float s_main,
                                                       float s_lo, int is_weight, unsigned short* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = src[i];
  const float l = x - (float)(half_t)x;
  const float b0 = is_weight ? x * s_main : l * s_lo;
  const float b1 = is_weight ? l * s_lo : x * s_main;
  out[i] = (unsigned short)__builtin_amdgcn_cvt_pk_fp8_f32(zk_clamp_fp8(b0), zk_clamp_fp8(b1), 0, false);
}

// activations [rows, K] fp32 -> row-scaled planes (zk_planes::rowexp): one wave per row, two passes over the row
__global__ __launch_bounds__(256) void split_rows_c8_kernel(const float* __restrict__ src, int rows, int K,
                                                            half_t* __restrict__ hi, half_t* __restrict__ c8,
                                                            int32_t* __restrict__ rowexp) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = src + (size_t)row * K;
  float amax = 0.f;
  for (int c = lane * 4; c < K; c += 256) {
    const f4_t v = *(const f4_t*)(xr + c);
    amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  const int sx = zk_row_exponent(amax);
  const float rs = ldexpf(1.0f, -sx);
  if (lane == 0) rowexp[row] = sx;
  for (int c = lane * 4; c < K; c += 256) {
    const f4_t v = *(const f4_t*)(xr + c);
    float y[4] = {v[0] * rs, v[1] * rs, v[2] * rs, v[3] * rs};
    h4_t h;
#pragma unroll
    for (int j = 0; j < 4; ++j) { zk_pin(y[j]); h[j] = (half_t)y[j]; }
    *(h4_t*)(hi + (size_t)row * K + c) = h;
    *(h4_t*)(c8 + (size_t)row * K + c) = zk_lo4(y, h, ZK_LO_C8);
  }
}

// columns [c0, c0 + ncols) of a [rows, ld] fp32 matrix -> activation c8 entries (test hooks / probes: the k columns of a
// fused QKV matrix as the ZK_F16C8 QKV epilogue writes them)
__global__ __launch_bounds__(256) void split_c8_cols_kernel(const float* __restrict__ src, int rows, int ld, int c0, int ncols,
                                                            half_t* __restrict__ lo) {
  const int per_row = ncols >> 2;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)rows * per_row) return;
  const size_t at = (size_t)(i / per_row) * ld + c0 + 4 * (int)(i % per_row);
  const f4_t v = *(const f4_t*)(src + at);
  const float y[4] = {v[0], v[1], v[2], v[3]};
  h4_t h;
#pragma unroll
  for (int j = 0; j < 4; ++j) h[j] = (half_t)y[j];
  *(h4_t*)(lo + at) = zk_lo4(y, h, ZK_LO_C8);
}

// WAV sample decode + channel mean (load_audio, src/test_long_audio_windows_2stage.py:54-56): interleaved little-endian
// samples -> one mono float32 per frame.  fmt: 1 = integer PCM (8 unsigned / 16 / 24 / 32 bit), 3 = IEEE float (32 / 64).
// Scaling as torchaudio.load(normalize=True): x / 2^(bits-1) (8-bit: (x-128)/128); channel mean = fp32 sum in channel
// order divided by the channel count (what wav.mean(dim=0) does for a handful of channels).
__global__ __launch_bounds__(256) void wav_decode_kernel(const unsigned char* __restrict__ raw, int64_t n_frames, int fmt,
                                                         int bits, int channels, float* __restrict__ out) {
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (f >= n_frames) return;
  const int bps = bits >> 3;
  const unsigned char* p = raw + (size_t)f * channels * bps;
  float acc = 0.f;
  for (int c = 0; c < channels; ++c, p += bps) {
    float v;
    if (fmt == 3) {
      if (bits == 32) { unsigned u = p[0] | (p[1] << 8) | (p[2] << 16) | ((unsigned)p[3] << 24); v = __uint_as_float(u); }
      else {
        unsigned long long u = 0;
        for (int b = 7; b >= 0; --b) u = (u << 8) | p[b];
        v = (float)__longlong_as_double((long long)u);
      }
    } else if (bits == 16) {
      v = (float)(short)(p[0] | (p[1] << 8)) * (1.0f / 32768.0f);
    } else if (bits == 8) {
      v = ((float)p[0] - 128.0f) * (1.0f / 128.0f);
    } else if (bits == 24) {
      int x = p[0] | (p[1] << 8) | (p[2] << 16);
      x = (x ^ 0x800000) - 0x800000;
      v = (float)x * (1.0f / 8388608.0f);
    } else {
      const int x = (int)(p[0] | (p[1] << 8) | (p[2] << 16) | ((unsigned)p[3] << 24));
      v = (float)((double)x / 2147483648.0);
    }
    acc = c == 0 ? v : acc + v;
  }
  out[f] = channels > 1 ? acc / (float)channels : acc;
}

// out[i*neu + p] = sum_j kernels[p][j] * padded[i*orig + j],  padded = zeros(width) ++ in ++ zeros(width+orig)
// (torchaudio.functional._apply_sinc_resample_kernel: conv1d with stride=orig over the padded waveform).
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ in, int64_t n_in, int orig, int neu,
                                                       int width, const float* __restrict__ kernels, int klen,
                                                       float* __restrict__ out, int64_t n_out, int span) {
  // A block's 256 consecutive outputs read one contiguous stretch of the input (frames i0 .. i1 of `orig` samples plus the
  // kernel's reach): it is staged ONCE into LDS with coalesced loads — the per-output reads are `orig` apart across the
  // lanes and overlap klen-fold — zeros outside the recording (the conv1d's padding).  The taps of the <= neu phases stay
  // in L1 (one broadcast address per phase).  Same accumulation order as before the staging: bit-identical outputs.
  extern __shared__ float tile[];
  const int64_t o0 = (int64_t)blockIdx.x * 256;
  const int64_t s0 = (o0 / neu) * orig - width;            // first input sample the block can touch
  for (int t = threadIdx.x; t < span; t += 256) {
    const int64_t sidx = s0 + t;
    tile[t] = (sidx >= 0 && sidx < n_in) ? in[sidx] : 0.f;
  }
  __syncthreads();
  const int64_t o = o0 + threadIdx.x;
  if (o >= n_out) return;
  const int64_t i = o / neu;
  const int p = (int)(o - i * neu);
  const float* kr = kernels + (size_t)p * klen;
  const float* x = tile + (int)(i * orig - width - s0);
  float acc = 0.f;
  for (int j = 0; j < klen; ++j) acc = fmaf(kr[j], x[j], acc);
  out[o] = acc;
}

}  // namespace

void zk_launch_split_f32(const float* src, int64_t n, float scale, half_t* hi, half_t* lo, hipStream_t s) {
  if (n <= 0) return;
  const int64_t thr = (n + 3) / 4;
  hipLaunchKernelGGL(split_kernel, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, s, src, n, scale, hi, lo);
}

void zk_launch_resample(const float* in, int64_t n_in, int orig, int neu, int width, const float* kernels, int klen,
                        float* out, int64_t n_out, hipStream_t s) {
  if (n_out <= 0) return;
  const int span = (255 / neu + 1) * orig + klen;      // input samples 256 consecutive outputs can reach
  hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), (size_t)span * 4, s, in, n_in, orig,
                     neu, width, kernels, klen, out, n_out, span);
}

void zk_launch_split_rows_c8(const float* src, int rows, int K, half_t* hi, half_t* c8, int32_t* rowexp, hipStream_t s) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(split_rows_c8_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, src, rows, K, hi, c8, rowexp);
}

void zk_launch_split_c8_cols(const float* src, int rows, int ld, int c0, int ncols, half_t* lo, hipStream_t s) {
  if (rows <= 0 || ncols <= 0) return;
  const long long n = (long long)rows * (ncols >> 2);
  hipLaunchKernelGGL(split_c8_cols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, rows, ld, c0, ncols, lo);
}

void zk_launch_split_c8(const float* src, int64_t n, int w_exp, int is_weight, half_t* c8, hipStream_t s) {
  if (n <= 0) return;
  const float s_main = is_weight ? ldexpf(1.0f, w_exp) : 1.0f;
  const float s_lo = ldexpf(1.0f, (is_weight ? w_exp : 0) + ZK_C8_SHIFT);
  hipLaunchKernelGGL(split_c8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, n, s_main, s_lo, is_weight,
                     (unsigned short*)c8);
}

void zk_launch_wav_decode(const unsigned char* raw, int64_t n_frames, int fmt, int bits, int channels, float* out,
                          hipStream_t s) {
  if (n_frames <= 0) return;
  hipLaunchKernelGGL(wav_decode_kernel, dim3((unsigned)((n_frames + 255) / 256)), dim3(256), 0, s, raw, n_frames, fmt, bits,
                     channels, out);
}
