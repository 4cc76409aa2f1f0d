// probe-only argument block of gemm_c6.hip: the ZK_F16C8 arguments with x_lo / w_lo pointing at MX-fp6 planes
// ([rows][K * 3 / 2] bytes: per 16 k-elements 32 e2m3 codes, (lo_i, value_i) interleaved for X, (value_i, lo_i) for W)
// plus the block-scale planes [K / 64][rows padded to 256][4] (e8m0; W's carry the 2^-11 of the split).
#pragma once
#include "zk_common.h"
struct zk_gemm6_args {
  zk_gemm_args a;
  const unsigned char* xs;
  const unsigned char* ws;
  int m_pad;      // rows of the X scale plane per k-step (M rounded up to 256)
};
void zk_launch_gemm_c6(const zk_gemm6_args& a, int epi, hipStream_t s);
