// Model tail: final LayerNorm of tokens 0/1 only, pooled = (cls + dist)/2, head LayerNorm, Linear(768 -> labels),
// plus the softmax / stage-1 gate used by the cascade.
// Replaces ASTModel.forward:302-304, ASTMLPHead.forward:315-318 and forward_probs' torch.softmax
// (src/test_long_audio_windows_2stage.py:111), and the gate at :312-320 / ..._cache.py:463-478.
// Only tokens 0 and 1 are consumed, so the final LayerNorm over the other 1212 tokens is never computed (exact).
#include "zk_common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// one wave per window; lane holds 12 channels (c = i*256 + lane*4 + j)
__global__ __launch_bounds__(64) void head_kernel(const float* __restrict__ hidden, int rows_per_window, int n_windows,
                                                  const float* __restrict__ lnf_g, const float* __restrict__ lnf_b,
                                                  const float* __restrict__ lnh_g, const float* __restrict__ lnh_b,
                                                  const float* __restrict__ w, const float* __restrict__ bias,
                                                  int num_labels, float eps, float* __restrict__ logits) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  if (b >= n_windows) return;
  float pooled[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) pooled[i] = 0.f;
  for (int tok = 0; tok < 2; ++tok) {
    const float* xr = hidden + ((size_t)b * rows_per_window + tok) * ZK_HIDDEN;
    float v[12];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const f4_t t = *(const f4_t*)(xr + i * 256 + lane * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[i * 4 + j] = t[j]; s += t[j]; }
    }
    const float mean = wave_sum(s) * (1.0f / ZK_HIDDEN);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) { const float d = v[i] - mean; q = fmaf(d, d, q); }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / ZK_HIDDEN) + eps);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = i * 256 + lane * 4 + j;
        pooled[i * 4 + j] += fmaf((v[i * 4 + j] - mean) * rstd, lnf_g[c], lnf_b[c]);
      }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 12; ++i) { pooled[i] *= 0.5f; s += pooled[i]; }
  const float mean = wave_sum(s) * (1.0f / ZK_HIDDEN);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 12; ++i) { const float d = pooled[i] - mean; q = fmaf(d, d, q); }
  const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / ZK_HIDDEN) + eps);
  float z[12];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = i * 256 + lane * 4 + j;
      z[i * 4 + j] = fmaf((pooled[i * 4 + j] - mean) * rstd, lnh_g[c], lnh_b[c]);
    }
  for (int o = 0; o < num_labels; ++o) {
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) d = fmaf(z[i * 4 + j], w[(size_t)o * ZK_HIDDEN + i * 256 + lane * 4 + j], d);
    d = wave_sum(d);
    if (lane == 0) logits[(size_t)b * num_labels + o] = d + bias[o];
  }
}

__global__ void softmax_kernel(const float* __restrict__ logits, int n, int num_labels, float* __restrict__ probs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* l = logits + (size_t)i * num_labels;
  float m = l[0];
  for (int j = 1; j < num_labels; ++j) m = fmaxf(m, l[j]);
  float s = 0.f;
  for (int j = 0; j < num_labels; ++j) s += expf(l[j] - m);
  for (int j = 0; j < num_labels; ++j) probs[(size_t)i * num_labels + j] = expf(l[j] - m) / s;
}

// stage-1 gate + ordered stream compaction, single workgroup (N is a few thousand windows at most per call):
// keep window i iff argmax == 1 (ties -> 0, as numpy argmax) and p_swallow >= thr1 [and >= fwd_min_prob if >= 0].
__global__ __launch_bounds__(1024) void gate_kernel(const float* __restrict__ logits, int n, float thr1,
                                                    float fwd_min_prob, float* __restrict__ probs,
                                                    int32_t* __restrict__ idx, int32_t* __restrict__ count) {
  __shared__ int wave_cnt[16];
  __shared__ int base;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) base = 0;
  __syncthreads();
  for (int start = 0; start < n; start += 1024) {
    const int i = start + tid;
    bool keep = false;
    if (i < n) {
      const float l0 = logits[2 * i], l1 = logits[2 * i + 1];
      const float m = fmaxf(l0, l1);
      const float e0 = expf(l0 - m), e1 = expf(l1 - m);
      const float s = e0 + e1;
      const float p0 = e0 / s, p1 = e1 / s;
      if (probs) { probs[2 * i] = p0; probs[2 * i + 1] = p1; }
      keep = (p1 > p0) && (p1 >= thr1) && (fwd_min_prob < 0.f || p1 >= fwd_min_prob);
    }
    const unsigned long long bal = __ballot(keep);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wv] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int k = 0; k < wv; ++k) off += wave_cnt[k];
    if (keep) idx[off + before] = i;
    __syncthreads();
    if (tid == 0) {
      int t = 0;
      for (int k = 0; k < 16; ++k) t += wave_cnt[k];
      base += t;
    }
    __syncthreads();
  }
  if (tid == 0) *count = base;
}

}  // namespace

void zk_launch_head(const float* hidden, int rows_per_window, int n_windows, const float* lnf_g, const float* lnf_b, const float* lnh_g,
                    const float* lnh_b, const float* w, const float* b, int num_labels, float eps, float* logits,
                    hipStream_t s) {
  if (n_windows <= 0) return;
  hipLaunchKernelGGL(head_kernel, dim3(n_windows), dim3(64), 0, s, hidden, rows_per_window, n_windows, lnf_g, lnf_b, lnh_g, lnh_b, w,
                     b, num_labels, eps, logits);
}

void zk_launch_softmax2(const float* logits, int n, int num_labels, float* probs, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(softmax_kernel, dim3((n + 255) / 256), dim3(256), 0, s, logits, n, num_labels, probs);
}

void zk_launch_gate(const float* logits, int n, float thr1, float fwd_min_prob, float* probs, int32_t* idx,
                    int32_t* count, hipStream_t s) {
  hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(1024), 0, s, logits, n, thr1, fwd_min_prob, probs, idx, count);
}
