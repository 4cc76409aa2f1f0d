// MFMA GEMM for the AST linear layers:  out[M,N] = X[M,K] · W[N,K]^T + bias  (+ fused epilogue)
//
// What it replaces: every nn.Linear of ASTAttention / ASTMLP and the patch-embedding Conv2d-as-GEMM
// ($TF/models/audio_spectrogram_transformer/modeling_audio_spectrogram_transformer.py:57-61,140-143,174,187-192).
//
// gfx950 design
//  * PERSISTENT: grid = min(#tiles, 256) workgroups of 512 threads (8 waves, one workgroup per CU); each workgroup
//    walks its list of output tiles and treats (tile, k-step) as ONE stream, so the global->LDS prefetch ring keeps
//    running across tile boundaries (no per-tile prologue bubble).  Tile order: consecutive groups of gridDim/8
//    tiles (n fastest) go to one XCD, so the workgroups of an XCD share X row panels in its private L2.
//  * W is the MFMA "A" operand (rows = n) and X the "B" operand (cols = m), v_mfma_f32_16x16x32_f16: a lane's 4
//    accumulator registers are 4 CONSECUTIVE n of one token row m -> 8 B (fp16) / 16 B (fp32) stores.
//  * tiles are staged HBM/L2 -> LDS with global_load_lds_dwordx4 into an NST-deep ring; the LDS image is linear per
//    wave-instruction, the 16-B chunk XOR swizzle f(row) = (row>>1)&(CPR-1) is applied to the SOURCE address and
//    again on the ds_read_b128 side (conflict-free for the 16x16x32 operand pattern, SQ_LDS_BANK_CONFLICT = 0).
//  * one raw s_barrier per k-step behind a COUNTED s_waitcnt vmcnt((NST-2)*loads_per_step): NST-2 later k-steps stay
//    in flight across the barrier (HBM/L2 latency under load is longer than one k-step of MFMAs).
//  * the bias slice of a tile rides the same ring (one global_load_lds_dword per wave at k = 0) so the epilogue
//    needs no VGPR-destination global load, which would make hipcc drain the ring with vmcnt(0).
//  * NSPLIT=3: every operand is an (hi, lo) fp16 pair, acc += Wl·Xh + Wh·Xl + Wh·Xh  (fp32-equivalent product,
//    2^-22 relative) — the mode that meets the 1e-3 logit tolerance; NSPLIT=1 is the plain fp16 pass.
#include "gemm_util.h"

namespace {

template <int NSPLIT, int BM, int BN, int BK, int WM, int WN, int NST, int EPI>
__global__ __launch_bounds__(512) void gemm_kernel(const zk_gemm_args a) {
  constexpr int NPL = (NSPLIT == 3) ? 2 : 1;
  constexpr int ROWB = BK * 2;               // bytes per tile row
  constexpr int CPR = ROWB / 16;             // 16-B chunks per row (8 or 4)
  constexpr int RPI = 1024 / ROWB;           // rows per wave-instruction (8 or 16)
  constexpr int TM = BM / WM, TN = BN / WN;  // per-wave extents
  constexpr int RM = TM / 16, RN = TN / 16;
  constexpr int KS = BK / 32;
  constexpr int XBYTES = BM * ROWB, WBYTES = BN * ROWB;
  constexpr int STAGE = NPL * (XBYTES + WBYTES);
  constexpr int XI = BM / RPI / 8, WI = BN / RPI / 8;   // glds per wave per plane
  constexpr int LPT = NPL * (XI + WI);                   // glds per wave per k-step
  constexpr int INFLIGHT = (NST - 2) * LPT;              // allowed to stay in flight across the barrier
  constexpr int BIAS_OFF = NST * STAGE;                  // 2 slots x 8 waves x 256 B
  constexpr int SCR_OFF = BIAS_OFF + 2 * 8 * 256;         // epilogue transpose scratch, per wave 16 rows x 144 B
  constexpr int SCR_STR = 144, SCR_WAVE = 16 * SCR_STR;
  static_assert(WM * WN == 8, "8 waves");
  static_assert(TN == 64, "bias slot is one dword per lane");
  static_assert((BM / RPI) % 8 == 0 && (BN / RPI) % 8 == 0, "staging split over 8 waves");
  static_assert(INFLIGHT < 64, "vmcnt field");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int tiles_n = a.N / BN;
  const int tiles_m = (a.M + BM - 1) / BM;
  const int ntiles = tiles_m * tiles_n;
  const int nk = a.K / BK;
  // tile list of this workgroup: T(s) = (s*8 + xcd)*per + j   (blocks b and b+8 share an XCD: speed only)
  const int per = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int first = xcd * per + jb;
  const int stride = 8 * per;
  const int my_tiles = first < ntiles ? (ntiles - first + stride - 1) / stride : 0;
  const int total = my_tiles * nk;

  const half_t* xp[2] = {a.x_hi, a.x_lo};
  const half_t* wp[2] = {a.w_hi, a.w_lo};

  // ---- staging: one k-step = LPT "pieces" (1-KiB global_load_lds_dwordx4 per wave), issued one per MFMA chunk ----
  const int srow = lane / CPR;               // row inside one wave-instruction
  const int schunk = lane % CPR;
  // load cursor (runs NST-1 steps ahead of the compute cursor)
  int l_step = 0, l_k = 0, l_ord = 0, l_tile = first;
  int l_m0 = (first / tiles_n) * BM, l_n0 = (first % tiles_n) * BN;
  // per-lane byte offsets of the pieces inside a tile's row panel: computed once (W) / once per tile (X, because of
  // the M-tail clamp), so that a piece is just  s_add base ; s_mov m0 ; global_load_lds v_off, s[base]  — no per-piece
  // vector address arithmetic in the k-loop (it was costing ~90 issue cycles per piece).
  unsigned xoffs[XI], woffs[WI];
  auto set_xoffs = [&]() {
#pragma unroll
    for (int q = 0; q < XI; ++q) {
      const int row = (q * 8 + wave) * RPI + srow;
      int grow = l_m0 + row;
      grow = grow < a.M ? grow : a.M - 1;
      const int c = schunk ^ ((row >> 1) & (CPR - 1));
      xoffs[q] = (unsigned)(grow - l_m0) * (unsigned)(a.K * 2) + (unsigned)(c * 16);
    }
  };
#pragma unroll
  for (int q = 0; q < WI; ++q) {
    const int row = (q * 8 + wave) * RPI + srow;
    const int c = schunk ^ ((row >> 1) & (CPR - 1));
    woffs[q] = (unsigned)row * (unsigned)(a.K * 2) + (unsigned)(c * 16);
  }
  set_xoffs();
  auto issue_piece = [&](int piece) {      // piece is a compile-time constant after unrolling
    if (l_step >= total) return;
    char* base = smem + (l_step % NST) * STAGE;
    const int p = piece / (XI + WI), q = piece % (XI + WI);
    const int k0 = l_k * BK;
    if (q < XI) {
      const int instr = q * 8 + wave;
      const char* gb = uniform_ptr((const char*)(xp[p] + (size_t)l_m0 * a.K + k0));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + xoffs[q]),
                                       (__attribute__((address_space(3))) void*)(base + p * XBYTES + instr * 1024),
                                       16, 0, 0);
    } else {
      const int instr = (q - XI) * 8 + wave;
      const char* gb = uniform_ptr((const char*)(wp[p] + (size_t)l_n0 * a.K + k0));
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(gb + woffs[q - XI]),
          (__attribute__((address_space(3))) void*)(base + NPL * XBYTES + p * WBYTES + instr * 1024), 16, 0, 0);
    }
    if (piece == 0 && l_k == 0) {
      const float* src = a.bias + l_n0 + wn * TN + lane;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(smem + BIAS_OFF +
                                                                                 ((l_ord & 1) * 8 + wave) * 256),
                                       4, 0, 0);
    }
  };
  auto advance_load = [&]() {
    if (l_step >= total) return;
    ++l_step;
    if (++l_k == nk) {
      l_k = 0; ++l_ord; l_tile += stride;
      const int tm = l_tile / tiles_n;
      l_m0 = tm * BM; l_n0 = (l_tile - tm * tiles_n) * BN;
      set_xoffs();
    }
  };
  auto issue = [&]() {
#pragma unroll
    for (int pc = 0; pc < LPT; ++pc) issue_piece(pc);
    advance_load();
  };

  f4_t acc[RN][RM];
#pragma unroll
  for (int i = 0; i < RN; ++i)
#pragma unroll
    for (int j = 0; j < RM; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

  // fragment addressing: row = tile_base + (lane&15); 16-B chunk = (lane>>4), XOR f(row) = (lane>>1)&(CPR-1)
  static_assert(KS == 1, "one 32-deep MFMA k-step per ring slot");
  const int frow = lane & 15;
  const int fsw = (lane >> 1) & (CPR - 1);
  const int fq = lane >> 4;
  const int xoff = (wm * TM + frow) * ROWB + ((fq ^ fsw) & (CPR - 1)) * 16;
  const int woff = (wn * TN + frow) * ROWB + ((fq ^ fsw) & (CPR - 1)) * 16;

  // The k-step is cut into NCH chunks of JB x NSPLIT MFMAs.  Each chunk first issues global->LDS pieces of the step
  // being prefetched and the LDS reads of the NEXT chunk's fragments, then its own MFMAs, so memory-instruction issue
  // and LDS latency sit in the shadow of the matrix pipe.  The LAST chunk of a step is DEFERRED past the step's
  // barrier (its fragments are already in registers): right after the barrier every wave has MFMAs to issue while
  // the first fragments of the new step are in flight — the barrier no longer drains the matrix pipe.
  constexpr int JB = (NSPLIT == 3) ? 4 : RM;
  constexpr int NJB = RM / JB;
  constexpr int NCH = NJB * RN;
  constexpr int XPC = (JB + RN - 1) / RN;          // next-block X fragments fetched per chunk
  h8_t xc_h[JB], xc_l[NPL == 2 ? JB : 1], xn_h[JB], xn_l[NPL == 2 ? JB : 1];
  h8_t wc_h, wc_l, wn_h, wn_l;

  auto load_x = [&](const char* xb, int jb, int j, h8_t& h, h8_t& l) {
    h = *(const h8_t*)(xb + xoff + (jb * JB + j) * 16 * ROWB);
    if constexpr (NPL == 2) l = *(const h8_t*)(xb + XBYTES + xoff + (jb * JB + j) * 16 * ROWB);
  };
  auto load_w = [&](const char* wb, int i, h8_t& h, h8_t& l) {
    h = *(const h8_t*)(wb + woff + i * 16 * ROWB);
    if constexpr (NPL == 2) l = *(const h8_t*)(wb + WBYTES + woff + i * 16 * ROWB);
  };
  // MFMAs of chunk c with up to JB LDS-DMA pieces of the prefetched step interleaved (one piece in front of each group
  // of NSPLIT MFMAs, so a piece's address arithmetic and issue overlap the matrix pipe instead of preceding it)
  auto mfma_chunk = [&](int c, int piece0, int npieces) {       // compile-time constants after unrolling
    const int jb = c / RN, i = c % RN;
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      if (j < npieces && piece0 + j < LPT) issue_piece(piece0 + j);
      if constexpr (NPL == 2) {
        acc[i][jb * JB + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc_l, xc_h[j], acc[i][jb * JB + j], 0, 0, 0);
        acc[i][jb * JB + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc_h, xc_l[j], acc[i][jb * JB + j], 0, 0, 0);
      }
      acc[i][jb * JB + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc_h, xc_h[j], acc[i][jb * JB + j], 0, 0, 0);
    }
  };
  constexpr int PCH = (LPT + JB - 1) / JB;               // chunks needed to carry all pieces (JB per chunk)
  static_assert(PCH <= NCH - 1, "pieces fit the non-deferred chunks");
  int c_ord = 0, c_tile = first;
  // ---- epilogue -------------------------------------------------------------------------------------------------
  // A lane's accumulator registers are 4 consecutive n of ONE row m (16 rows per register tile): stored directly that
  // is 32-B pieces of 16 different rows per instruction, which is store-issue bound.  Each 16-row block is therefore
  // transposed through a 2.3 KB per-wave LDS scratch so that every global instruction moves 8 whole 128-B rows
  // (16 B per lane), for the fp16 planes as well as for the fp32 residual read-modify-write.
  char* scr = smem + SCR_OFF + wave * SCR_WAVE;
  const int rd_row = lane >> 3, rd_ch = lane & 7;
  auto epilogue = [&]() {
    const int tm = c_tile / tiles_n, tn = c_tile - tm * tiles_n;
    const int m0 = tm * BM + wm * TM, n0 = tn * BN + wn * TN;
    const char* bslot = smem + BIAS_OFF + ((c_ord & 1) * 8 + wave) * 256;
    f4_t b4[RN];
#pragma unroll
    for (int i = 0; i < RN; ++i) b4[i] = *(const f4_t*)(bslot + (i * 16 + 4 * fq) * 4);
    if constexpr (EPI == ZK_EPI_PATCH) {
#pragma unroll
      for (int i = 0; i < RN; ++i)
#pragma unroll
        for (int j = 0; j < RM; ++j) {
          const int m = m0 + j * 16 + frow, n = n0 + i * 16 + 4 * fq;
          f4_t v = acc[i][j] + b4[i];
          acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};
          // pin v in front of the divergent tail guard: hipcc otherwise sinks the MFMA that produces acc[i][j] into
          // the guarded block, where it would run with a partial EXEC mask (wrong A/B rows from the masked lanes)
          asm volatile("" : "+v"(v));
          if (m >= a.M) continue;
          // A row m -> (window b, patch p = f·101 + t); with patch_tr the rows hold the time patches t < patch_tr only
          const int npr = a.patch_tr ? ZK_FOUT * a.patch_tr : ZK_NPATCH;
          const int b = m / npr;
          int p = m - b * npr;
          if (a.patch_tr) { const int f = p / a.patch_tr; p = f * ZK_TOUT + (p - f * a.patch_tr); }
          const f4_t pe = *(const f4_t*)(a.pos + (size_t)(p + 2) * a.N + n);
          *(f4_t*)(a.resid + ((size_t)b * ZK_SEQ + 2 + p) * a.N + n) = v + pe;
        }
    } else if constexpr (EPI == ZK_EPI_RESID) {
      static_assert(RN == 4, "two halves of two 16-column blocks");
#pragma unroll
      for (int j = 0; j < RM; ++j)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
          for (int il = 0; il < 2; ++il) {
            *(f4_t*)(scr + frow * SCR_STR + il * 64 + fq * 16) = acc[hf * 2 + il][j] + b4[hf * 2 + il];
            acc[hf * 2 + il][j] = f4_t{0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const f4_t v = *(const f4_t*)(scr + (rd_row + 8 * t) * SCR_STR + rd_ch * 16);
            const int m = m0 + j * 16 + rd_row + 8 * t;
            if (m < a.M) {
              float* dst = a.resid + (size_t)m * a.N + n0 + hf * 32 + rd_ch * 4;
              *(f4_t*)dst = *(const f4_t*)dst + v;
            }
          }
        }
    } else {
      const bool want_lo = a.o_lo != nullptr && n0 < a.lo_n_limit;
      [[maybe_unused]] const gelu_coef_t gk = gelu_coefficients();
#pragma unroll
      for (int j = 0; j < RM; ++j) {
        h4_t lo4[RN];
#pragma unroll
        for (int i = 0; i < RN; ++i) {
          f4_t v = acc[i][j] + b4[i];
          acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};
          if constexpr (EPI == ZK_EPI_GELU) {
            const gelu_f2_t g01 = gelu_erf2(gelu_f2_t{v[0], v[1]}, gk), g23 = gelu_erf2(gelu_f2_t{v[2], v[3]}, gk);
            v[0] = g01[0]; v[1] = g01[1]; v[2] = g23[0]; v[3] = g23[1];
          }
          h4_t hi;
#pragma unroll
          for (int e = 0; e < 4; ++e) { hi[e] = (half_t)v[e]; lo4[i][e] = (half_t)(v[e] - (float)hi[e]); }
          *(h4_t*)(scr + frow * SCR_STR + i * 32 + fq * 8) = hi;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const h8_t v = *(const h8_t*)(scr + (rd_row + 8 * t) * SCR_STR + rd_ch * 16);
          const int m = m0 + j * 16 + rd_row + 8 * t;
          if (m < a.M) *(h8_t*)(a.o_hi + (size_t)m * a.ldo + n0 + rd_ch * 8) = v;
        }
        if (want_lo) {
#pragma unroll
          for (int i = 0; i < RN; ++i) *(h4_t*)(scr + frow * SCR_STR + i * 32 + fq * 8) = lo4[i];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const h8_t v = *(const h8_t*)(scr + (rd_row + 8 * t) * SCR_STR + rd_ch * 16);
            const int m = m0 + j * 16 + rd_row + 8 * t;
            if (m < a.M) *(h8_t*)(a.o_lo + (size_t)m * a.ldo + n0 + rd_ch * 8) = v;
          }
        }
      }
    }
    ++c_ord; c_tile += stride;
  };

  if (total == 0) return;
#pragma unroll
  for (int i = 0; i < NST - 1; ++i)
    if (l_step < total) issue();
  // first step landed?  (conservative: drain; happens once per launch)
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  int c_k = 0;
  bool epi_pending = false;
  for (int c_step = 0; c_step < total; ++c_step) {
    const char* xb = smem + (c_step % NST) * STAGE;
    const char* wb = xb + NPL * XBYTES;
    // ---- slot 0: first fragments of this step in flight behind the deferred chunk of the previous step ----
#pragma unroll
    for (int j = 0; j < JB; ++j) load_x(xb, 0, j, xn_h[j], xn_l[j]);
    load_w(wb, 0, wn_h, wn_l);
    if (c_step > 0) {
      mfma_chunk(NCH - 1, 0, 0);
      if (epi_pending) {
        epilogue();
        epi_pending = false;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < JB; ++j) { xc_h[j] = xn_h[j]; if constexpr (NPL == 2) xc_l[j] = xn_l[j]; }
    wc_h = wn_h;
    if constexpr (NPL == 2) wc_l = wn_l;
    // ---- chunks 0 .. NCH-2 ----
#pragma unroll
    for (int c = 0; c < NCH - 1; ++c) {
      const int jb = c / RN, i = c % RN;
      load_w(wb, (c + 1) % RN, wn_h, wn_l);
      if (jb + 1 < NJB) {
#pragma unroll
        for (int q = 0; q < XPC; ++q)
          if (i * XPC + q < JB) load_x(xb, jb + 1, i * XPC + q, xn_h[i * XPC + q], xn_l[i * XPC + q]);
      }
      if (c < PCH) mfma_chunk(c, c * JB, JB);
      else mfma_chunk(c, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      wc_h = wn_h;
      if constexpr (NPL == 2) wc_l = wn_l;
      if (i == RN - 1 && jb + 1 < NJB) {
#pragma unroll
        for (int j = 0; j < JB; ++j) { xc_h[j] = xn_h[j]; if constexpr (NPL == 2) xc_l[j] = xn_l[j]; }
      }
    }
    advance_load();
    if (++c_k == nk) { c_k = 0; epi_pending = true; }
    // ---- step c_step+1 must have landed; NST-2 younger steps may stay in flight.  xc/wc now hold the fragments of
    //      the deferred chunk NCH-1 (read from this slot BEFORE the barrier that frees it) ----
    if (l_step - c_step - 2 >= NST - 2) wait_vmcnt<INFLIGHT>();
    else wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  mfma_chunk(NCH - 1, 0, 0);
  epilogue();
}

template <int NSPLIT, int EPI>
void launch_cfg(const zk_gemm_args& a, hipStream_t s) {
  constexpr int BM = 256, BN = 256, BK = 32;
  constexpr int NST = (NSPLIT == 3) ? 2 : 4;
  constexpr int NPL = (NSPLIT == 3) ? 2 : 1;
  constexpr int lds = NST * NPL * (BM + BN) * BK * 2 + 2 * 8 * 256 + 8 * 16 * 144;
  auto k = gemm_kernel<NSPLIT, BM, BN, BK, 2, 4, NST, EPI>;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
  const int ntiles = ((a.M + BM - 1) / BM) * (a.N / BN);
  int grid = ntiles < 256 ? ((ntiles + 7) / 8) * 8 : 256;
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, s, a);
}

template <int NSPLIT>
void launch_epi(const zk_gemm_args& a, int epi, hipStream_t s) {
  switch (epi) {
    case ZK_EPI_STORE: launch_cfg<NSPLIT, ZK_EPI_STORE>(a, s); break;
    case ZK_EPI_GELU: launch_cfg<NSPLIT, ZK_EPI_GELU>(a, s); break;
    case ZK_EPI_RESID: launch_cfg<NSPLIT, ZK_EPI_RESID>(a, s); break;
    default: launch_cfg<NSPLIT, ZK_EPI_PATCH>(a, s); break;
  }
}

}  // namespace

// Host-side shape contract (checked by the caller, zkast.hip): N % 256 == 0, K % 64 == 0, M >= 1, all planes
// 16-byte aligned, x planes hold at least M rows.
void zk_launch_gemm(const zk_gemm_args& a_in, int epi, int nsplit, hipStream_t s) {
  zk_gemm_args a = a_in;
  if (a.ldo == 0) a.ldo = a.N;
  if (nsplit == 3) launch_epi<3>(a, epi, s);
  else launch_epi<1>(a, epi, s);
}
