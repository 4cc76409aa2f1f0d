// Device helpers shared by the MFMA GEMM kernels (gemm.hip, gemm_c8.hip).
#pragma once
#include "zk_common.h"

namespace {

// exact-erf GELU ($TF/activations.py:70-89) as  max(x, 0) - |x|·s(|x|),  s(u) = 0.5·erfc(u/sqrt 2) = 2^-P(u):
// P is a degree-8 fit of -log2(0.5·erfc(u/sqrt 2)) on [0, 8] (weighted by u·s, the sensitivity of the result), so the
// whole function is 8 fma + one v_exp_f32 + 4 plain ops — no reciprocal, no sign fix-up, and no cancellation for
// negative x (the small values keep their relative accuracy).  |error| <= 2.5e-7 (half an ulp of the result near x = 4.7),
// relative error of the negative tail <= 8e-6; beyond |x| = 8 the tail term is < 1e-14.
__device__ __forceinline__ float gelu_erf(float x) {
  const float u = fminf(fabsf(x), 8.0f);
  float p = 1.690407657e-06f;
  p = fmaf(p, u, -2.508334364e-05f);
  p = fmaf(p, u, 1.144626513e-04f);
  p = fmaf(p, u, 3.233428288e-04f);
  p = fmaf(p, u, -7.333386224e-03f);
  p = fmaf(p, u, 5.271420255e-02f);
  p = fmaf(p, u, 4.591154456e-01f);
  p = fmaf(p, u, 1.151123285e+00f);
  p = fmaf(p, u, 9.999988675e-01f);
  const float s = __builtin_amdgcn_exp2f(-p);
  return fmaxf(x, 0.0f) - u * s;
}

// two elements at a time: the Horner chain as v_pk_fma_f32 (hipcc turns the scalar form above into one v_fmaak_f32 per
// coefficient and element, twice the issue slots)
typedef float gelu_f2_t __attribute__((ext_vector_type(2)));
#ifndef ZK_GELU_DEG
#define ZK_GELU_DEG 8
#endif
struct gelu_coef_t { float c[ZK_GELU_DEG + 1]; };
// the coefficients as opaque scalar registers, fetched once per epilogue (an empty asm hides the literals: folded into the
// instructions they would turn every packed fma back into two v_fmaak_f32)
__device__ __forceinline__ gelu_coef_t gelu_coefficients() {
#if ZK_GELU_DEG == 6      // probe: degree-6 fit (|error| 1.3e-7, relative error of the negative tail 5e-4 up to |x| = 5)
  gelu_coef_t k = {{9.999952316e-01f, 1.151190996e+00f, 4.587764442e-01f, 5.343326181e-02f, -8.108191192e-03f,
                    7.806957001e-04f, -3.466758062e-05f}};
#else
  gelu_coef_t k = {{9.999988675e-01f, 1.151123285e+00f, 4.591154456e-01f, 5.271420255e-02f, -7.333386224e-03f,
                    3.233428288e-04f, 1.144626513e-04f, -2.508334364e-05f, 1.690407657e-06f}};
#endif
#pragma unroll
  for (int i = 0; i <= ZK_GELU_DEG; ++i) asm volatile("" : "+s"(k.c[i]));
  return k;
}
__device__ __forceinline__ gelu_f2_t gelu_erf2(gelu_f2_t x, const gelu_coef_t& k) {
  const gelu_f2_t u = {fminf(fabsf(x[0]), 8.0f), fminf(fabsf(x[1]), 8.0f)};
  gelu_f2_t p = {k.c[ZK_GELU_DEG], k.c[ZK_GELU_DEG]};
#pragma unroll
  for (int i = ZK_GELU_DEG - 1; i >= 0; --i) p = __builtin_elementwise_fma(p, u, gelu_f2_t{k.c[i], k.c[i]});
  const gelu_f2_t s2 = {__builtin_amdgcn_exp2f(-p[0]), __builtin_amdgcn_exp2f(-p[1])};
  const gelu_f2_t r = {fmaxf(x[0], 0.0f), fmaxf(x[1], 0.0f)};
  return r - u * s2;
}

// the round-1 formulation (Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7 on erf), kept for the frozen A/B kernel
__device__ __forceinline__ float gelu_erf_as26(float x) {
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
