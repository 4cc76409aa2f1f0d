// Kernel probes for tools/ — NOT part of libzkast.so (build.sh links this file only into libzkast_probes.so, which
// tools/ scripts load through ZKAST_LIB): device-resident GEMM timing of the production launchers with interleaved
// A/B rounds in one process, and a bit-exact comparison of two variants' outputs at production sizes.
#include "zk_common.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

void zk_launch_gemm_c8_v1(const zk_gemm_args& a, int epi, hipStream_t s);

namespace {

__device__ __forceinline__ unsigned mix32(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return (unsigned)x;
}
// approximately normal(0, sigma) from 4 uniform bytes (full sign/exponent variety matters for the DVFS clock)
__global__ __launch_bounds__(256) void fill_kernel(float* out, int64_t n, unsigned seed, float sigma) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const unsigned u = mix32((unsigned long long)i * 0x9E3779B97F4A7C15ULL + seed);
  const float s = (float)(u & 255) + (float)((u >> 8) & 255) + (float)((u >> 16) & 255) + (float)(u >> 24);
  out[i] = (s - 510.0f) * (sigma / 147.8f);
}
__global__ __launch_bounds__(256) void diff_kernel(const unsigned* a, const unsigned* b, int64_t n, unsigned long long* cnt) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n && a[i] != b[i]) atomicAdd(cnt, 1ULL);
}


// ---- fill probe: the GEMM's LDS-DMA stream alone (same tile walk, same pieces: 8 per wave and 64-KiB step), no MFMA, no
// ds_read.  depth = steps in flight (1 or 2; LDS ring of depth+... slots of 64 KiB), fix: 1 = X always row block 0,
// 2 = W always column tile 0.  Answers: what the fill path of a CU delivers on this access pattern.
template <int DEPTH>
__global__ __launch_bounds__(512) void fill_probe_kernel(const half_t* x_hi, const half_t* x_lo, const half_t* w_hi,
                                                         const half_t* w_lo, int M, int N, int K, int fix) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tiles_n = N / 256, tiles_m = (M + 255) / 256, ntiles = tiles_m * tiles_n, nk = K / 64;
  const int per = gridDim.x >> 3;
  const int first = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  const int stride = 8 * per;
  const int srow = lane / 8, schunk = lane % 8;
  unsigned poffs[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = (q * 8 + wave) * 8 + srow;
    poffs[q] = (unsigned)row * (unsigned)(K * 2) + (unsigned)((schunk ^ ((row >> 1) & 7)) * 16);
  }
  int slot = 0;
  for (int tile = first; tile < ntiles; tile += stride) {
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = (fix & 1) ? 0 : tm * 256, n0 = (fix & 2) ? 0 : tn * 256;
    for (int k = 0; k < nk; ++k)
      for (int kind = 0; kind < 2; ++kind) {
        const half_t* xp = kind ? x_lo : x_hi;
        const half_t* wp = kind ? w_lo : w_hi;
        const char* xb = (const char*)(xp + (size_t)m0 * K + k * 64);
        const char* wb = (const char*)(wp + (size_t)n0 * K + k * 64);
        char* base = smem + slot * 65536;
        // fix & 4: the X plane as [row block][k step][256 rows][128 B] tiles -> a step's X is 32 KiB CONTIGUOUS
        const char* xt = (const char*)xp + ((size_t)(m0 / 256) * nk + k) * 32768;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((fix & 4) ? xt + (q * 8 + wave) * 1024 + lane * 16 : xb + poffs[q]),
                                           (__attribute__((address_space(3))) void*)(base + (q * 8 + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wb + poffs[q]),
                                           (__attribute__((address_space(3))) void*)(base + 32768 + (q * 8 + wave) * 1024), 16, 0, 0);
        slot = slot + 1 == DEPTH + 0 ? 0 : slot + 1;
        if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
  }
}

#define CK(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { fprintf(stderr, "probe: %s failed: %s\n", #expr, hipGetErrorString(e__)); return -2; } } while (0)

int c8_exp(float mx) {
  if (!(mx > 0.f)) return 0;
  return (int)floorf(log2f(224.0f / mx));
}

}  // namespace

extern "C" {

// Times zk_launch_gemm_c8 (variant 1 = the first-round kernel, 2 = ping-pong) on random device-resident operands.
// variants: bit mask (1 | 2); rounds interleave the variants; ms_out[v] = median over rounds of the average launch
// time of `iters` back-to-back launches.  mismatch_out: dwords of the outputs that differ between the variants.
int zkp_bench_gemm_c8(int M, int N, int K, int epi, int variants, int iters, int rounds, float* ms_out,
                      unsigned long long* mismatch_out) {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  // x planes hold whole 256-row tiles (the kernels read rows >= M of the last tile and never store them)
  // ZKP_TILED=1: the launch form the forward uses (zk_planes::tiled: X read as k-slice-major tiles, the GELU epilogue writes
  // them) on the same random buffers — a permutation of random data is random data, so timing, clock and traffic are those of
  // the production launch (the values are not compared with a row-major result)
  const bool tiled = getenv("ZKP_TILED") != nullptr;
  const size_t nx = (size_t)((M + 255) / 256 * 256) * K, nw = (size_t)N * K, no = (size_t)(tiled ? (M + 255) / 256 * 256 : M) * N;
  float *fx, *fw, *bias, *resid[2] = {nullptr, nullptr};
  half_t *xh, *xl, *wh, *wl, *oh[2] = {nullptr, nullptr}, *ol[2] = {nullptr, nullptr};
  CK(hipMalloc((void**)&fx, nx * 4)); CK(hipMalloc((void**)&fw, nw * 4)); CK(hipMalloc((void**)&bias, (size_t)N * 4));
  CK(hipMalloc((void**)&xh, nx * 2)); CK(hipMalloc((void**)&xl, nx * 2));
  CK(hipMalloc((void**)&wh, nw * 2)); CK(hipMalloc((void**)&wl, nw * 2));
  // ZKP_ZERO=1: all-zero operands (the DVFS check of MI355X_MICROARCH.md "give-back": same cycles, higher clock)
  const float zx = getenv("ZKP_ZERO") ? 0.0f : 1.0f;
  hipLaunchKernelGGL(fill_kernel, dim3((nx + 255) / 256), dim3(256), 0, s, fx, (int64_t)nx, 1u, 1.0f * zx);
  hipLaunchKernelGGL(fill_kernel, dim3((nw + 255) / 256), dim3(256), 0, s, fw, (int64_t)nw, 2u, 0.05f * zx);
  hipLaunchKernelGGL(fill_kernel, dim3((N + 255) / 256), dim3(256), 0, s, bias, (int64_t)N, 3u, 0.1f);
  const int w_exp = c8_exp(0.05f * 3.45f);
  zk_launch_split_f32(fx, (int64_t)nx, 1.f, xh, nullptr, s);
  zk_launch_split_f32(fw, (int64_t)nw, 1.f, wh, nullptr, s);
  zk_launch_split_c8(fx, (int64_t)nx, 0, 0, xl, s);
  zk_launch_split_c8(fw, (int64_t)nw, w_exp, 1, wl, s);
  CK(hipStreamSynchronize(s));
  (void)hipFree(fx); (void)hipFree(fw);
  const bool rmw = epi == ZK_EPI_RESID;
  for (int v = 0; v < 2; ++v) {
    if (!(variants & (1 << v))) continue;
    if (rmw) { CK(hipMalloc((void**)&resid[v], no * 4)); CK(hipMemsetAsync(resid[v], 0, no * 4, s)); }
    else { CK(hipMalloc((void**)&oh[v], no * 2)); CK(hipMalloc((void**)&ol[v], no * 2)); CK(hipMemsetAsync(oh[v], 0, no * 2, s)); CK(hipMemsetAsync(ol[v], 0, no * 2, s)); }
  }
  auto args = [&](int v) {
    zk_gemm_args a;
    a.x_hi = xh; a.x_lo = xl; a.w_hi = wh; a.w_lo = wl; a.bias = bias; a.x_rowexp = nullptr; a.M = M; a.N = N; a.K = K; a.x_rows = (M + 255) / 256 * 256;
    a.o_hi = oh[v]; a.o_lo = ol[v]; a.resid = resid[v]; a.pos = nullptr; a.lo_n_limit = N; a.lo_c8_to = epi == ZK_EPI_STORE ? (2 * N) / 3 : 1 << 30;
    a.w_exp = w_exp; a.lo_c8_from = epi == ZK_EPI_STORE ? N / 3 : 1 << 30;
    a.x_tiled = tiled ? 1 : 0; a.o_tiled = (tiled && epi == ZK_EPI_GELU) ? 1 : 0;
    return a;
  };
  auto launch = [&](int v) {
    const zk_gemm_args a = args(v);
    if (v == 0) zk_launch_gemm_c8_v1(a, epi, s); else zk_launch_gemm_c8(a, epi, s);
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> t[2];
  for (int r = 0; r < rounds + 1; ++r)
    for (int v = 0; v < 2; ++v) {
      if (!(variants & (1 << v))) continue;
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < iters; ++i) launch(v);
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) t[v].push_back(ms / iters);      // round 0 = warm-up
    }
  CK(hipGetLastError());
  for (int v = 0; v < 2; ++v) {
    ms_out[v] = 0.f;
    if (t[v].empty()) continue;
    std::sort(t[v].begin(), t[v].end());
    ms_out[v] = t[v][t[v].size() / 2];
  }
  *mismatch_out = 0;
  if (variants == 3) {      // one clean launch each (RESID accumulates, so reset first), then compare every output dword
    unsigned long long* cnt;
    CK(hipMalloc((void**)&cnt, 8)); CK(hipMemsetAsync(cnt, 0, 8, s));
    for (int v = 0; v < 2; ++v) { if (rmw) CK(hipMemsetAsync(resid[v], 0, no * 4, s)); launch(v); }
    if (rmw) hipLaunchKernelGGL(diff_kernel, dim3((no + 255) / 256), dim3(256), 0, s, (const unsigned*)resid[0], (const unsigned*)resid[1], (int64_t)no, cnt);
    else {
      hipLaunchKernelGGL(diff_kernel, dim3((no / 2 + 255) / 256), dim3(256), 0, s, (const unsigned*)oh[0], (const unsigned*)oh[1], (int64_t)(no / 2), cnt);
      hipLaunchKernelGGL(diff_kernel, dim3((no / 2 + 255) / 256), dim3(256), 0, s, (const unsigned*)ol[0], (const unsigned*)ol[1], (int64_t)(no / 2), cnt);
    }
    CK(hipMemcpyAsync(mismatch_out, cnt, 8, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    (void)hipFree(cnt);
  }
  CK(hipStreamSynchronize(s));
  for (void* p : {(void*)bias, (void*)xh, (void*)xl, (void*)wh, (void*)wl, (void*)oh[0], (void*)oh[1], (void*)ol[0], (void*)ol[1], (void*)resid[0], (void*)resid[1]})
    if (p) (void)hipFree(p);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipStreamDestroy(s);
  return 0;
}

// returns ms of one launch (median of rounds) of the fill probe
int zkp_fill_probe(int M, int N, int K, int depth, int fix, int rounds, float* ms_out) {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const size_t nx = (size_t)((M + 255) / 256 * 256) * K, nw = (size_t)N * K;
  half_t *xh, *xl, *wh, *wl;
  CK(hipMalloc((void**)&xh, nx * 2)); CK(hipMalloc((void**)&xl, nx * 2));
  CK(hipMalloc((void**)&wh, nw * 2)); CK(hipMalloc((void**)&wl, nw * 2));
  CK(hipMemsetAsync(xh, 1, nx * 2, s)); CK(hipMemsetAsync(xl, 2, nx * 2, s));
  CK(hipMemsetAsync(wh, 3, nw * 2, s)); CK(hipMemsetAsync(wl, 4, nw * 2, s));
  const int lds = depth * 65536;
  auto k1 = fill_probe_kernel<1>; auto k2 = fill_probe_kernel<2>;
  (void)hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  (void)hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> t;
  for (int r = 0; r < rounds + 1; ++r) {
    CK(hipEventRecord(e0, s));
    if (depth == 2) hipLaunchKernelGGL(k2, dim3(256), dim3(512), lds, s, xh, xl, wh, wl, M, N, K, fix);
    else hipLaunchKernelGGL(k1, dim3(256), dim3(512), lds, s, xh, xl, wh, wl, M, N, K, fix);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0) t.push_back(ms);
  }
  CK(hipGetLastError());
  std::sort(t.begin(), t.end());
  *ms_out = t[t.size() / 2];
  (void)hipFree(xh); (void)hipFree(xl); (void)hipFree(wh); (void)hipFree(wl);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(s);
  return 0;
}

// attention alone on random device-resident planes: ms per launch (median of rounds), nsplit 3 (split QK^T), 2 (fp8-corrected QK^T) or 1
int zkp_bench_attention(int n_windows, int nsplit, int iters, int rounds, float* ms_out) {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const size_t rows = (size_t)n_windows * ZK_SEQ, nq = rows * 3 * ZK_HIDDEN, no = rows * ZK_HIDDEN;
  float* f;
  half_t *qh, *ql, *oh, *ol;
  CK(hipMalloc((void**)&f, nq * 4)); CK(hipMalloc((void**)&qh, nq * 2)); CK(hipMalloc((void**)&ql, nq * 2));
  CK(hipMalloc((void**)&oh, no * 2)); CK(hipMalloc((void**)&ol, no * 2));
  hipLaunchKernelGGL(fill_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, f, (int64_t)nq, 7u, 1.0f);
  zk_launch_split_f32(f, (int64_t)nq, 1.f, qh, ql, s);
  if (nsplit == 2) zk_launch_split_c8_cols(f, (int)rows, 3 * ZK_HIDDEN, ZK_HIDDEN, ZK_HIDDEN, ql, s);
  CK(hipStreamSynchronize(s));
  (void)hipFree(f);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> t;
  for (int r = 0; r < rounds + 1; ++r) {
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i)
      zk_launch_attention(zk_planes{qh, nsplit >= 2 ? ql : nullptr, ZK_LO_F16, nullptr}, zk_planes{oh, ol, ZK_LO_C8, nullptr}, n_windows, nsplit, 0, s);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0) t.push_back(ms / iters);
  }
  CK(hipGetLastError());
  std::sort(t.begin(), t.end());
  *ms_out = t[t.size() / 2];
  (void)hipFree(qh); (void)hipFree(ql); (void)hipFree(oh); (void)hipFree(ol);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(s);
  return 0;
}

}  // extern "C"
