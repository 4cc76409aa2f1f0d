// Probe for the MX-fp6 correction-plane GEMM variant (gemm_c6.hip) — libzkast_probes.so only.
// zkp_bench_gemm_c6 builds both operand forms from the same random fp32 matrices, times the production ZK_F16C8 kernel and
// the fp6 variant in interleaved rounds and reports how far their outputs are apart (both approximate the same fp32
// product: the difference is the sum of their correction-term errors).
#include "gemm_c6.h"

#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace {

typedef float f16v_t __attribute__((ext_vector_type(16)));
typedef unsigned u6v_t __attribute__((ext_vector_type(6)));

__device__ __forceinline__ unsigned mix32b(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return (unsigned)x;
}
__global__ __launch_bounds__(256) void fill6_kernel(float* out, int64_t n, unsigned seed, float sigma) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const unsigned u = mix32b((unsigned long long)i * 0x9E3779B97F4A7C15ULL + seed);
  const float s = (float)(u & 255) + (float)((u >> 8) & 255) + (float)((u >> 16) & 255) + (float)(u >> 24);
  out[i] = (s - 510.0f) * (sigma / 147.8f);
}

// one thread per (row, block of 16 k): 32 e2m3 codes = (a_i, b_i) interleaved, a = lo·2^11 / b = value for activations,
// a = value / b = lo·2^11 for weights; one e8m0 scale per block chosen so that the block maximum lands in (3.75, 7.75)
// (the conversion saturates at 7.5); the weights' stored scale carries the 2^-11 of the split.
__global__ __launch_bounds__(256) void pack_c6_kernel(const float* __restrict__ src, int rows, int K, int rows_pad, int is_w,
                                                      unsigned char* __restrict__ plane, unsigned char* __restrict__ scales) {
  const int nblk = K / 16;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)rows_pad * nblk) return;
  const int row = (int)(idx / nblk), blk = (int)(idx % nblk);
  f16v_t va, vb;
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float v = row < rows ? src[(size_t)row * K + blk * 16 + i] : 0.f;
    const float lo = (v - (float)(half_t)v) * 2048.f;
    va[i] = is_w ? v : lo;
    vb[i] = is_w ? lo : v;
    amax = fmaxf(amax, fmaxf(fabsf(v), fabsf(lo)));
  }
  int sb = (int)(((__float_as_uint(amax) + 0x80000u) >> 23) & 0xff) - 2;      // mantissa >= 15/16 rounds up to the next 8
  sb = sb < 12 ? 12 : sb;
  const u6v_t code = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(va, vb, __uint_as_float((unsigned)sb << 23));
  unsigned* dst = (unsigned*)(plane + (size_t)row * (K * 3 / 2) + blk * 24);
#pragma unroll
  for (int i = 0; i < 6; ++i) dst[i] = code[i];
  scales[((size_t)(blk >> 2) * rows_pad + row) * 4 + (blk & 3)] = (unsigned char)(is_w ? sb - ZK_C8_SHIFT : sb);
}

__global__ __launch_bounds__(256) void maxdiff_kernel(const void* a, const void* b, int64_t n, int is_half, unsigned* out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = is_half ? (float)((const half_t*)a)[i] : ((const float*)a)[i];
  const float y = is_half ? (float)((const half_t*)b)[i] : ((const float*)b)[i];
  atomicMax(out, __float_as_uint(fabsf(x - y)));
  atomicMax(out + 1, __float_as_uint(fabsf(x)));
}

#define CK(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { fprintf(stderr, "probe: %s failed: %s\n", #expr, hipGetErrorString(e__)); return -2; } } while (0)

}  // namespace

extern "C" {

// ms_out[0] = production ZK_F16C8 kernel, ms_out[1] = fp6 variant (median over rounds of the average of `iters` launches);
// err_out[0] = max |out6 - out8|, err_out[1] = max |out8|
int zkp_bench_gemm_c6(int M, int N, int K, int epi, int iters, int rounds, float* ms_out, float* err_out) {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int m_pad = (M + 255) / 256 * 256;
  const size_t nx = (size_t)m_pad * K, nw = (size_t)N * K, no = (size_t)M * N;
  float *fx, *fw, *bias, *resid[2] = {nullptr, nullptr};
  half_t *xh, *xl, *wh, *wl, *oh[2] = {nullptr, nullptr}, *ol[2] = {nullptr, nullptr};
  unsigned char *x6, *w6, *xs, *ws;
  CK(hipMalloc((void**)&fx, nx * 4)); CK(hipMalloc((void**)&fw, nw * 4)); CK(hipMalloc((void**)&bias, (size_t)N * 4));
  CK(hipMalloc((void**)&xh, nx * 2)); CK(hipMalloc((void**)&xl, nx * 2));
  CK(hipMalloc((void**)&wh, nw * 2)); CK(hipMalloc((void**)&wl, nw * 2));
  CK(hipMalloc((void**)&x6, nx * 3 / 2)); CK(hipMalloc((void**)&w6, nw * 3 / 2));
  CK(hipMalloc((void**)&xs, nx / 16)); CK(hipMalloc((void**)&ws, nw / 16));
  hipLaunchKernelGGL(fill6_kernel, dim3((nx + 255) / 256), dim3(256), 0, s, fx, (int64_t)nx, 1u, 1.0f);
  hipLaunchKernelGGL(fill6_kernel, dim3((nw + 255) / 256), dim3(256), 0, s, fw, (int64_t)nw, 2u, 0.05f);
  hipLaunchKernelGGL(fill6_kernel, dim3((N + 255) / 256), dim3(256), 0, s, bias, (int64_t)N, 3u, 0.1f);
  const int w_exp = (int)floorf(log2f(224.0f / (0.05f * 3.45f)));
  zk_launch_split_f32(fx, (int64_t)nx, 1.f, xh, nullptr, s);
  zk_launch_split_f32(fw, (int64_t)nw, 1.f, wh, nullptr, s);
  zk_launch_split_c8(fx, (int64_t)nx, 0, 0, xl, s);
  zk_launch_split_c8(fw, (int64_t)nw, w_exp, 1, wl, s);
  hipLaunchKernelGGL(pack_c6_kernel, dim3((unsigned)(((size_t)m_pad * (K / 16) + 255) / 256)), dim3(256), 0, s, fx, m_pad, K, m_pad, 0, x6, xs);
  hipLaunchKernelGGL(pack_c6_kernel, dim3((unsigned)(((size_t)N * (K / 16) + 255) / 256)), dim3(256), 0, s, fw, N, K, N, 1, w6, ws);
  CK(hipStreamSynchronize(s));
  CK(hipGetLastError());
  (void)hipFree(fx); (void)hipFree(fw);
  const bool rmw = epi == ZK_EPI_RESID;
  for (int v = 0; v < 2; ++v) {
    if (rmw) { CK(hipMalloc((void**)&resid[v], no * 4)); CK(hipMemsetAsync(resid[v], 0, no * 4, s)); }
    else { CK(hipMalloc((void**)&oh[v], no * 2)); CK(hipMalloc((void**)&ol[v], no * 2)); CK(hipMemsetAsync(oh[v], 0, no * 2, s)); CK(hipMemsetAsync(ol[v], 0, no * 2, s)); }
  }
  auto launch = [&](int v) {
    zk_gemm6_args g;
    zk_gemm_args& a = g.a;
    a.x_hi = xh; a.w_hi = wh; a.bias = bias; a.x_rowexp = nullptr; a.M = M; a.N = N; a.K = K; a.x_rows = (M + 255) / 256 * 256;
    a.o_hi = oh[v]; a.o_lo = ol[v]; a.resid = resid[v]; a.pos = nullptr; a.lo_n_limit = epi == ZK_EPI_STORE ? (2 * N) / 3 : N;
    a.w_exp = w_exp; a.lo_c8_from = epi == ZK_EPI_STORE ? N / 3 : 1 << 30;
    if (v == 0) { a.x_lo = xl; a.w_lo = wl; zk_launch_gemm_c8(a, epi, s); }
    else { a.x_lo = (const half_t*)x6; a.w_lo = (const half_t*)w6; g.xs = xs; g.ws = ws; g.m_pad = m_pad; zk_launch_gemm_c6(g, epi, s); }
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> t[2];
  for (int r = 0; r < rounds + 1; ++r)
    for (int v = 0; v < 2; ++v) {
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < iters; ++i) launch(v);
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) t[v].push_back(ms / iters);
    }
  CK(hipGetLastError());
  for (int v = 0; v < 2; ++v) { std::sort(t[v].begin(), t[v].end()); ms_out[v] = t[v][t[v].size() / 2]; }
  unsigned* d;
  CK(hipMalloc((void**)&d, 8)); CK(hipMemsetAsync(d, 0, 8, s));
  for (int v = 0; v < 2; ++v) { if (rmw) CK(hipMemsetAsync(resid[v], 0, no * 4, s)); launch(v); }
  if (rmw) hipLaunchKernelGGL(maxdiff_kernel, dim3((unsigned)((no + 255) / 256)), dim3(256), 0, s, (const void*)resid[1], (const void*)resid[0], (int64_t)no, 0, d);
  else hipLaunchKernelGGL(maxdiff_kernel, dim3((unsigned)((no + 255) / 256)), dim3(256), 0, s, (const void*)oh[1], (const void*)oh[0], (int64_t)no, 1, d);
  unsigned hd[2];
  CK(hipMemcpyAsync(hd, d, 8, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  // maxdiff writes |x - y| to [0] and |x| (the variant's) to [1]; report the reference magnitude from the same slot
  memcpy(&err_out[0], &hd[0], 4); memcpy(&err_out[1], &hd[1], 4);
  for (void* p : {(void*)d, (void*)bias, (void*)xh, (void*)xl, (void*)wh, (void*)wl, (void*)x6, (void*)w6, (void*)xs, (void*)ws, (void*)oh[0], (void*)oh[1],
                  (void*)ol[0], (void*)ol[1], (void*)resid[0], (void*)resid[1]})
    if (p) (void)hipFree(p);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(s);
  return 0;
}

}  // extern "C"
