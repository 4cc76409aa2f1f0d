// Device helpers shared by the MFMA GEMM kernels (gemm.hip, gemm_c8.hip).
#pragma once
#include "zk_common.h"

namespace {

__device__ __forceinline__ float gelu_erf(float x) {
  // 0.5 x (1 + erf(x / sqrt 2)); erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7), exact-erf GELU as
  // $TF/activations.py:70-89 to well below the fp32 noise of the surrounding GEMMs.
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p *= t;
  const float e = __expf(-z * z);
  const float erf_abs = fmaf(-p, e, 1.0f);
  const float erf_v = copysignf(erf_abs, x);
  return 0.5f * x * (1.0f + erf_v);
}

// make wave-uniformity of a pointer provable to the compiler, so that  ptr + zext(u32 lane offset)  selects the
// SGPR-base + VGPR-offset addressing mode
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
  const unsigned long long g = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)g);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(g >> 32));
  return (const char*)(((unsigned long long)hi << 32) | lo);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace
