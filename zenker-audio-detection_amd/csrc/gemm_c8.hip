// MFMA GEMM, "fp16 + fp8-corrected" mode (ZK_F16C8):  out[M,N] = X[M,K] · W[N,K]^T + bias  (+ fused epilogue)
//
// Same role as gemm.hip (every nn.Linear of ASTAttention / ASTMLP and the patch-embedding Conv2d-as-GEMM,
// $TF/models/audio_spectrogram_transformer/modeling_audio_spectrogram_transformer.py:57-61,140-143,174,187-192) and the
// same fp32-grade product, at 2 matrix-pipe passes instead of 3:
//
//     X·W = Xh·Wh + (Xl·W + X·Wl) + O(2^-22),        Xh = fp16(X), Xl = X - Xh  (|Xl| <= 2^-11 |X|), same for W.
//
// The bracket is 2^-11 of the result, so 4 significant bits are plenty for it.  Each operand therefore carries, next
// to its fp16 plane, a "c8" plane of byte pairs  X' = (fp8(Xl·2^11), fp8(X)),  W' = (fp8(W·2^e), fp8(Wl·2^(e+11)))
// (OCP e4m3).  Read as rows of 2K bytes, X'·W'^T IS the bracket times 2^(e+11): one fp8 GEMM with K' = 2K on
// v_mfma_scale_f32_16x16x128_f8f6f4, whose A-side block scale 2^-(e+11) puts it straight into the accumulator of
// the fp16 product (2x the fp16 rate -> the K' = 2K product costs one fp16 pass).
//
// gfx950 structure (differences to gemm.hip):
//  * a ring step is 64 k-elements = 128-byte rows for BOTH plane kinds (fp16: 64 x 2 B, c8: 64 x 2 B), 64 KiB per
//    step, two ring slots.  Steps alternate  main(k) -> slot 0,  corr(k) -> slot 1,  so slot, plane and MFMA kind of
//    a step are compile-time constants of a 2x-unrolled loop.
//  * a lane's fragment of a 16-row tile is 32 bytes of its row: 16-B chunks q and q+4 (q = lane>>4) of the
//    XOR-swizzled row image (conflict-free ds_read_b128 pair).  Both operands use the same K permutation, which is
//    all an MFMA needs; the fp16 step feeds the two halves to two 16x16x32 MFMAs, the fp8 step feeds all 8 VGPRs
//    to one 16x16x128 MFMA.
//  * W-stationary: the 4 W fragments of the wave's 64 columns stay in registers for the whole step, the 8 X
//    fragments stream through a 2-deep register buffer (24 ds_read_b128 per step and wave).  The last X tile of a
//    step is multiplied AFTER the step's barrier, behind the first fragment reads of the next step.
#include <type_traits>

#include "gemm_util.h"

#ifdef ZK_C8_STAMPS
// probe builds only (tools/gemm_stamps.py): three s_memtime stamps per ring step and wave — A = every MFMA and piece of the
// step is issued, B = the next step's data has landed (vmcnt(0)), C = the step barrier has released — written with scalar
// stores into a ring of the last 64 steps per (workgroup, wave).  (Cprev -> A = issue phase, A -> B = data wait, B -> C =
// waiting for the other waves.)  The stamps cost three SMEM round trips per step; nothing else in the kernel changes.
__device__ unsigned zk_c8_stamp_buf[256 * 8 * 64 * 4];
// per workgroup: s_memtime and s_memrealtime (100 MHz) at kernel entry and exit -> the clock the chip held under this kernel
// (MI355X guide, DVFS give-back item 6: clock = d(memtime) / d(memrealtime) x 100 MHz)
__device__ unsigned long long zk_c8_clock_buf[256 * 4];
extern "C" int zkp_c8_stamps_read(unsigned* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(zk_c8_stamp_buf), sizeof(zk_c8_stamp_buf));
}
extern "C" int zkp_c8_clock_read(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(zk_c8_clock_buf), sizeof(zk_c8_clock_buf));
}
#endif

namespace {

// symmetric form (ZK_C8_ROLES=0): the 8 LDS-DMA pieces a wave issues per step: 2 in the deferred tile right behind the step barrier (chunk -1), 2 in
// each of chunks 0..2 (measured best of four placements; a piece costs ~100 issue cycles among MFMAs)
constexpr int c8_cnt(int c) {
  constexpr int T[8] = {2, 2, 2, 2, 0, 0, 0, 0};
  return T[c + 1];
}
constexpr int c8_base(int c) { int b = 0; for (int q = -1; q < c; ++q) b += c8_cnt(q); return b; }
// piece to issue in front of MFMA group g (of G) of chunk c, or -1
constexpr int c8_piece(int c, int g, int G) {
  const int n = c8_cnt(c);
  for (int q = 0; q < n; ++q) if (q * G / n == g) return c8_base(c) + q;
  return -1;
}

// Roles (ZK_C8_ROLES = piece table, 0 = the symmetric round-1/2 form above; ZK_C8_RLOAD = which half loads): the two waves
// of a SIMD take different roles — the LOADER wave issues all 16 LDS-DMA pieces of its SIMD per step, its partner issues
// none, so one MFMA stream per SIMD never stalls at a vector-memory issue.  Measured (profiles/r03_gemm_roles_ab.txt,
// bit-identical outputs): loaders = the OLDER waves 0-3 +1.5…2.9 % on all four GEMM shapes, loaders = waves 4-7 −4…−10 %
// (the younger wave loses the matrix-pipe arbitration anyway; stalled at pieces on top of that it becomes the straggler
// of every step).  Table = pieces per chunk (-1 .. 6); front-loaded 4-4-4-4 is the production one.
#ifndef ZK_C8_ROLES
#define ZK_C8_ROLES 1
#endif
#ifndef ZK_C8_RLOAD
#define ZK_C8_RLOAD 0
#endif
#if ZK_C8_ROLES
constexpr int c8r_cnt(int c) {
#if ZK_C8_ROLES == 1
  constexpr int T[8] = {4, 4, 4, 4, 0, 0, 0, 0};
#elif ZK_C8_ROLES == 2
  constexpr int T[8] = {2, 2, 2, 2, 2, 2, 2, 2};
#elif ZK_C8_ROLES == 3
  constexpr int T[8] = {16, 0, 0, 0, 0, 0, 0, 0};
#elif ZK_C8_ROLES == 4
  constexpr int T[8] = {8, 8, 0, 0, 0, 0, 0, 0};
#elif ZK_C8_ROLES == 6
  constexpr int T[8] = {6, 6, 4, 0, 0, 0, 0, 0};
#elif ZK_C8_ROLES == 7
  constexpr int T[8] = {4, 4, 4, 2, 2, 0, 0, 0};
#else
  constexpr int T[8] = {4, 3, 3, 2, 2, 2, 0, 0};
#endif
  return T[c + 1];
}
constexpr int c8r_base(int c) { int b = 0; for (int q = -1; q < c; ++q) b += c8r_cnt(q); return b; }
#endif

// compile-time loop: f(integral_constant<int, I>) for I in [0, N) — the loop index ends up in "n" asm operands
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

typedef int i4v_t __attribute__((ext_vector_type(4)));
typedef int i8v_t __attribute__((ext_vector_type(8)));
struct frag_t { i4v_t a, b; };

// XT / OT (zk_gemm_args::x_tiled / o_tiled, zk_planes::tiled): activation planes in k-slice-major tiles — [row block][64-k
// chunk][256 rows][128 B], the 128 bytes of a row already in the LDS image's swizzled chunk order — so that the X half of a
// ring step is 32 KiB CONTIGUOUS in HBM (one DRAM page run instead of 256 row segments 1.5-6 KiB apart) and an LDS-DMA piece
// is a linear kilobyte.  XT: the loader reads X that way; OT (GELU epilogue): the output planes are written that way.
template <int EPI, bool XT, bool OT>
__global__ __launch_bounds__(512) void gemm_c8_kernel(const zk_gemm_args a) {
  static_assert(!XT || ZK_C8_ROLES, "the tiled operand form is read by the role loader only");
  static_assert(!OT || EPI == ZK_EPI_GELU, "only the GELU epilogue writes tiled planes");
  constexpr int BM = 256, BN = 256, BK = 64, WM = 2, WN = 4;
  constexpr int ROWB = 128, CPR = 8, RPI = 8;
  constexpr int TM = BM / WM, TN = BN / WN;       // 128 x 64 per wave
  constexpr int RM = TM / 16, RN = TN / 16;       // 8 X tiles, 4 W tiles
  constexpr int XBYTES = BM * ROWB, WBYTES = BN * ROWB, STAGE = XBYTES + WBYTES;
  constexpr int XI = BM / RPI / 8, WI = BN / RPI / 8;
  [[maybe_unused]] constexpr int LPT = XI + WI;   // 1-KiB LDS-DMA pieces per wave and step
  constexpr int BIAS_OFF = 2 * STAGE;
  constexpr int SCR_OFF = BIAS_OFF + 2 * 8 * 256;
  constexpr int SCR_STR = 144, SCR_WAVE = 16 * SCR_STR;
  constexpr int RSC_OFF = SCR_OFF + 8 * SCR_WAVE;      // per wave: 128 row scales (fp32) of its X rows, epilogue only
  constexpr int TILE_B = 16 * ROWB;               // 2048 B between consecutive 16-row tiles

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int tiles_n = a.N / BN;
  const int tiles_m = (a.M + BM - 1) / BM;
  const int nk = a.K / BK;
  const int per = gridDim.x >> 3;
#ifndef ZK_C8_WALK
#define ZK_C8_WALK 0
#endif
#if ZK_C8_WALK
  // probe (round 5, VERDICT item 3: what W's re-fetch costs in clock): the 32 workgroups of an XCD form one SUPER TILE of
  // WR row blocks x WC column tiles, and the super tiles are walked column-group-major, so all eight XCDs stay on the same WC
  // column panels of W (WC x 786 KB for K = 768: L2-resident) for tiles_m / (8 WR) consecutive rounds; an X row block is
  // shared by the WC workgroups of ONE XCD and read again once per column group.  Tile index t = 32 s + j of the linear
  // walk (s = super tile, j = workgroup of the XCD) is all `phys` needs.  Row blocks past tiles_m (the last row group is
  // partial) load the last real block and store nothing (m >= M).
  constexpr int WC = ZK_C8_WALK, WR = 32 / WC;
  const bool walk = (EPI == ZK_EPI_GELU || EPI == ZK_EPI_STORE) && tiles_n % WC == 0 && per == 32 && !a.rev && tiles_m >= 8 * WR;
  const int n_rg = (tiles_m + WR - 1) / WR;
  const int ntiles = walk ? n_rg * WR * tiles_n : tiles_m * tiles_n;
#else
  const int ntiles = tiles_m * tiles_n;
#endif
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int first = xcd * per + jb;
  const int stride = 8 * per;
  const int my_tiles = first < ntiles ? (ntiles - first + stride - 1) / stride : 0;
  const int total = my_tiles * nk * 2;
  if (total == 0) return;
#ifdef ZK_C8_STAMPS
  {
    unsigned long long t0, r0;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) : : "memory");
    unsigned long long* cp = zk_c8_clock_buf + blockIdx.x * 4;
    asm volatile("s_store_dwordx2 %1, %0, 0x0\n\ts_store_dwordx2 %2, %0, 0x8" : : "s"(cp), "s"(t0), "s"(r0) : "memory");
  }
#endif
#ifndef ZK_C8_STAGGER
#define ZK_C8_STAGGER 12
#endif

  // RESID (O projection, FC2): de-phase the XCDs.  All tiles of a launch take the same time, so without this every CU
  // reaches its epilogue — a burst of fp32 residual reads and writes, matrix pipe idle — at the same moment and the
  // bursts queue at HBM.  XCD x starts x·nk·ZK_C8_STAGGER·64 cycles late (up to about half a tile time); the workgroups
  // of one XCD stay in step (they share X panels through their L2).  Measured +2-5 % (O) / +4-6 % (FC2) at 28 tiles per
  // workgroup; the store-only epilogues (QKV, FC1) lose 2-3 % with it and keep the common start, and launches of a few
  // tiles per workgroup would only pay the late finish (17-window micro-batch: 0.7x), hence the tile-count guard.
  if constexpr (EPI == ZK_EPI_RESID)
    if (my_tiles >= 8)
      for (int i = 0; i < xcd * nk; ++i) __builtin_amdgcn_s_sleep(ZK_C8_STAGGER);
#ifdef ZK_C8_STAGGER_STORE      // probe: the same for the store-only epilogues
  if constexpr (EPI == ZK_EPI_STORE || EPI == ZK_EPI_GELU)
    if (my_tiles >= 8)
      for (int i = 0; i < xcd * nk; ++i) __builtin_amdgcn_s_sleep(ZK_C8_STAGGER_STORE);
#endif

  // ---- staging ----
  const int srow = lane / CPR, schunk = lane % CPR;
  // a.rev: the tile walk runs from the LAST row block to the first (tile t of the walk is tile ntiles-1-t of the matrix): the
  // kernel then starts on the rows its producer wrote last, which may still sit in the 256 MiB Infinity Cache (zkast.hip
  // alternates the direction along the kernel chain)
#if ZK_C8_WALK
  auto phys = [&](int t) __attribute__((always_inline)) {
    if (!walk) return a.rev ? ntiles - 1 - t : t;
    const int sp = t >> 5, j = t & 31, cg = sp / n_rg, rg = sp - cg * n_rg;
    return (rg * WR + j / WC) * tiles_n + cg * WC + j % WC;
  };
  auto ld_row = [&](int tm) __attribute__((always_inline)) { return (tm < tiles_m ? tm : tiles_m - 1) * BM; };
#else
  auto phys = [&](int t) __attribute__((always_inline)) { return a.rev ? ntiles - 1 - t : t; };
  auto ld_row = [&](int tm) __attribute__((always_inline)) { return tm * BM; };
#endif
  int l_step = 0, l_k = 0, l_ord = 0, l_tile = first;
  int l_m0 = ld_row(phys(first) / tiles_n), l_n0 = (phys(first) % tiles_n) * BN;
  // ONE per-lane offset serves every piece: piece q of either operand covers rows q·64 + wave·8 + srow of the tile, and
  // the swizzled chunk ((row >> 1) & 7 does not depend on q), so q moves into the wave-uniform base address.  (Rows >= M
  // of the last row block are read as they lie — the planes hold whole 256-row tiles, zk_gemm_args — and never stored.)
  unsigned poff;
  {
#if ZK_C8_ROLES
    const int row = (wave & 3) * RPI + srow;      // loader lw = wave & 3 covers 8-row units lw and lw + 4 of every 64 rows
#else
    const int row = wave * RPI + srow;
#endif
    const int c = schunk ^ ((row >> 1) & (CPR - 1));
    poff = (unsigned)row * (unsigned)(a.K * 2) + (unsigned)(c * 16);
  }
  // piece `pc` of the step at the load cursor; KIND (0 = fp16 planes -> slot 0, 1 = c8 planes -> slot 1) is static.
  // Pieces are UNCONDITIONAL so that a ring step is one basic block (hipcc otherwise sinks the MFMAs of a step below
  // all of its memory instructions): once the cursor has run past the last step it stays on it and the pieces
  // re-fetch that step into the slot nobody reads any more.
  auto issue_piece = [&](auto kind_c, int pc) __attribute__((always_inline)) {
    constexpr int KIND = decltype(kind_c)::value;
    char* base = smem + KIND * STAGE;
    const half_t* xpl = KIND ? a.x_lo : a.x_hi;
    const half_t* wpl = KIND ? a.w_lo : a.w_hi;
    const int k0 = l_k * BK;
    if (pc < XI) {
      const int instr = pc * 8 + wave;
      const char* gb = uniform_ptr((const char*)(xpl + (size_t)(l_m0 + pc * 8 * RPI) * a.K + k0));
      asm volatile("" : "+v"(poff));      // keep the offset a 32-bit VGPR: SGPR-base + voffset addressing
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + poff),
                                       (__attribute__((address_space(3))) void*)(base + instr * 1024), 16, 0, 0);
    } else {
      const int instr = (pc - XI) * 8 + wave;
      const char* gb = uniform_ptr((const char*)(wpl + (size_t)(l_n0 + (pc - XI) * 8 * RPI) * a.K + k0));
      asm volatile("" : "+v"(poff));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + poff),
                                       (__attribute__((address_space(3))) void*)(base + XBYTES + instr * 1024), 16, 0,
                                       0);
    }
    if (KIND == 0 && pc == 0) {      // the tile's bias slice rides along with every fp16 step (256 B per wave)
      // (lane offset from a volatile v_mbcnt: a hoisted 64-bit per-lane pointer would cost two VGPRs across the k-loop)
      unsigned l4;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 2, %0" : "=v"(l4));
      const char* src = uniform_ptr((const char*)(a.bias + l_n0 + wn * TN)) + l4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(smem + BIAS_OFF +
                                                                                 ((l_ord & 1) * 8 + wave) * 256),
                                       4, 0, 0);
    }
  };
#ifndef ZK_C8_BUFLDS
#define ZK_C8_BUFLDS 0
#endif
#if ZK_C8_BUFLDS
  const __amdgpu_buffer_rsrc_t rs_xh = __builtin_amdgcn_make_buffer_rsrc((void*)a.x_hi, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_xl = __builtin_amdgcn_make_buffer_rsrc((void*)a.x_lo, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wh = __builtin_amdgcn_make_buffer_rsrc((void*)a.w_hi, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wl = __builtin_amdgcn_make_buffer_rsrc((void*)a.w_lo, 0, -1, 0x00020000);
#endif
#if ZK_C8_ROLES
  // loader-role piece p of 16: X pieces 0..7, W pieces 8..15; piece (q, h) = (p >> 1 & 3, p & 1) covers the 8-row unit
  // q·8 + lw + 4h of its operand (rows q·64 + h·32 + lw·8 + srow: the swizzle term (row >> 1) & 7 depends on lw, srow only)
  auto issue_piece_r = [&](auto kind_c, int p) __attribute__((always_inline)) {
    constexpr int KIND = decltype(kind_c)::value;
    char* base = smem + KIND * STAGE;
    const half_t* xpl = KIND ? a.x_lo : a.x_hi;
    const half_t* wpl = KIND ? a.w_lo : a.w_hi;
    const int k0 = l_k * BK;
    const int q = (p >> 1) & 3, h = p & 1, unit = q * 8 + (wave & 3) + 4 * h;
    const bool isw = p >= 8;
#if ZK_C8_BUFLDS
    // probe (round 4): the same pieces as `buffer_load_dwordx4 ... offen lds` — one resource per plane held in SGPRs for
    // the whole kernel, the piece's position as a 32-bit scalar offset (no 64-bit pointer arithmetic, no v_readfirstlane
    // pair per piece), the lane's part as the 32-bit vector offset
    if (XT && !isw) {
      unsigned l16;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 4, %0" : "=v"(l16));
      const unsigned so = ((unsigned)((l_m0 >> 8) * nk + l_k) << 15) + (unsigned)(unit * 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(KIND ? rs_xl : rs_xh, (__attribute__((address_space(3))) void*)(base + unit * 1024), 16,
                                               (int)l16, (int)so, 0, 0);
    } else {
      const int r0 = (isw ? l_n0 : l_m0) + q * 64 + h * 32;
      const unsigned so = ((unsigned)r0 * (unsigned)a.K + (unsigned)k0) * 2u;
      asm volatile("" : "+v"(poff));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(isw ? (KIND ? rs_wl : rs_wh) : (KIND ? rs_xl : rs_xh),
                                               (__attribute__((address_space(3))) void*)(base + (isw ? XBYTES : 0) + unit * 1024), 16,
                                               (int)poff, (int)so, 0, 0);
    }
#else
    if (XT && !isw) {      // (p is a constant at every call site: no branch is emitted)
      unsigned l16;      // lane·16, read here instead of kept in a register across the k-loop
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 4, %0" : "=v"(l16));
      const char* gt = uniform_ptr((const char*)xpl + (((size_t)(l_m0 >> 8) * nk + l_k) << 15) + unit * 1024);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gt + l16),
                                       (__attribute__((address_space(3))) void*)(base + unit * 1024), 16, 0, 0);
    } else {
    const half_t* pl = isw ? wpl : xpl;
    const int r0 = (isw ? l_n0 : l_m0) + q * 64 + h * 32;
    const char* gb = uniform_ptr((const char*)(pl + (size_t)r0 * a.K + k0));
    asm volatile("" : "+v"(poff));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + poff),
                                     (__attribute__((address_space(3))) void*)(base + (isw ? XBYTES : 0) + unit * 1024), 16, 0, 0);
    }
#endif
    if (KIND == 0 && p == 0) {
      unsigned l4;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 2, %0" : "=v"(l4));
      const char* src = uniform_ptr((const char*)(a.bias + l_n0 + wn * TN)) + l4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(smem + BIAS_OFF +
                                                                                 ((l_ord & 1) * 8 + (wave & 3)) * 256),
                                       4, 0, 0);
    }
  };
#endif
  auto advance_load = [&](auto kind_c) __attribute__((always_inline)) {
    constexpr int KIND = decltype(kind_c)::value;
    ++l_step;
    if (KIND == 1 && l_step < total) {      // past the end the cursor stays on the last step (see issue_piece)
      if (++l_k == nk) {
        l_k = 0; ++l_ord; l_tile += stride;
        const int pt = phys(l_tile), tm = pt / tiles_n;
        l_m0 = ld_row(tm); l_n0 = (pt - tm * tiles_n) * BN;
      }
    }
  };


  f4_t acc[RN][RM];
#pragma unroll
  for (int i = 0; i < RN; ++i)
#pragma unroll
    for (int j = 0; j < RM; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

  // fragment addressing: row = tile_base + (lane&15), chunks q and q+4 of the swizzled row
  const int frow = lane & 15;
  const int fq = lane >> 4;
  const int fsw = (lane >> 1) & 7;
  const int xoff = (wm * TM + frow) * ROWB + ((fq ^ fsw) & 7) * 16;
  const int woff = XBYTES + (wn * TN + frow) * ROWB + ((fq ^ fsw) & 7) * 16;
  // Fragment reads are inline asm with HAND-COUNTED s_waitcnt lgkmcnt: hipcc's own placement waits lgkmcnt(0) in front
  // of a chunk's first MFMA, i.e. also for the two reads of the NEXT tile issued just before — an exposed LDS round
  // trip per chunk.  LDS reads retire in order, so "all but the newest two" is lgkmcnt(2).  The wait statements name
  // the fragments they make valid as in/out operands so that no MFMA can be scheduled in front of them.
  const unsigned lds0 = (unsigned)(uintptr_t)smem;      // LDS byte address of the ring (low half of the flat address)
  const unsigned xad[2] = {lds0 + (unsigned)xoff, lds0 + (unsigned)(xoff ^ 64)};
  const unsigned wad[2] = {lds0 + (unsigned)woff, lds0 + (unsigned)(woff ^ 64)};
  auto ld_frag = [&](frag_t& f, auto slot_c, const unsigned (&ad)[2], auto tile_c) __attribute__((always_inline)) {
    constexpr int IMM = decltype(tile_c)::value * TILE_B;
    constexpr unsigned SLOT = decltype(slot_c)::value * STAGE;
    asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b128 %1, %3 offset:%4"
                 : "=&v"(f.a), "=&v"(f.b)
                 : "v"(ad[0] + SLOT), "v"(ad[1] + SLOT), "n"(IMM));
  };
  // e8m0 block scales of the fp8 MFMA: W' side 2^-(w_exp + 11), X' side 1
  const int sc_w = (127 - (a.w_exp + ZK_C8_SHIFT)) * 0x01010101;
  const int sc_x = 0x7f7f7f7f;
  // HALF: fp16 kind: 0 / 1 = first / second 32-deep MFMA of the 64-deep step (issued as two sweeps over the W tiles, so
  // consecutive MFMAs never depend on each other); fp8 kind: one 128-byte-deep MFMA (HALF 0 only)
  auto mma = [&](auto kind_c, auto half_c, f4_t& c, const frag_t& w, const frag_t& x) __attribute__((always_inline)) {
    constexpr int KIND = decltype(kind_c)::value;
    constexpr int HALF = decltype(half_c)::value;
    if constexpr (KIND == 0) {
      if constexpr (HALF == 0)
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, w.a), __builtin_bit_cast(h8_t, x.a), c, 0, 0, 0);
      else
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, w.b), __builtin_bit_cast(h8_t, x.b), c, 0, 0, 0);
    } else if constexpr (HALF == 0) {
      const i8v_t wv = __builtin_shufflevector(w.a, w.b, 0, 1, 2, 3, 4, 5, 6, 7);
      const i8v_t xv = __builtin_shufflevector(x.a, x.b, 0, 1, 2, 3, 4, 5, 6, 7);
      c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wv, xv, c, 0, 0, 0, sc_w, 0, sc_x);
    }
  };

  frag_t wf[RN];      // W tiles of the current step
  frag_t xf[2];       // X tile j lives in xf[j & 1]

  int c_ord = 0, c_tile = first;
  // ---- epilogue (accumulator layout and LDS-transposed stores as in gemm.hip) ----
  char* scr = smem + SCR_OFF + wave * SCR_WAVE;
  const int rd_row = lane >> 3, rd_ch = lane & 7;
  auto epilogue = [&]() __attribute__((always_inline)) {
    const int pt = phys(c_tile), tm = pt / tiles_n, tn = pt - tm * tiles_n;
    const int m0 = tm * BM + wm * TM, n0 = tn * BN + wn * TN;
#if ZK_C8_ROLES
    const char* bslot = smem + BIAS_OFF + ((c_ord & 1) * 8 + (wave & 3)) * 256;
#else
    const char* bslot = smem + BIAS_OFF + ((c_ord & 1) * 8 + wave) * 256;
#endif
    f4_t b4[RN];
#pragma unroll
    for (int i = 0; i < RN; ++i) b4[i] = *(const f4_t*)(bslot + (i * 16 + 4 * fq) * 4);
    // row-scaled X planes (zk_planes::rowexp): accumulator row m is multiplied by 2^s_m — folded into the bias add as
    // one fma per element.  The wave's 128 scales go through LDS so that one VGPR at a time holds them.
    const bool scaled = a.x_rowexp != nullptr;
    float* rsc = (float*)(smem + RSC_OFF + wave * 512);
    if (scaled) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        int m = m0 + t * 64 + lane;
        m = m < a.M ? m : a.M - 1;
        rsc[t * 64 + lane] = __int_as_float((a.x_rowexp[m] + 127) << 23);
      }
    }
    auto row_scale = [&](int j) __attribute__((always_inline)) { return scaled ? rsc[j * 16 + frow] : 1.0f; };
#ifdef ZK_C8_NOEPI      // probe builds only (tools/build_variant.sh): the tile's results are dropped -> time of the k-loops alone
    if (a.M > 0) {
#pragma unroll
      for (int i = 0; i < RN; ++i)
#pragma unroll
        for (int j = 0; j < RM; ++j) { asm volatile("" : "+v"(acc[i][j])); acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f}; }
    } else
#endif
    if constexpr (EPI == ZK_EPI_PATCH) {
#pragma unroll
      for (int i = 0; i < RN; ++i)
#pragma unroll
        for (int j = 0; j < RM; ++j) {
          const int m = m0 + j * 16 + frow, n = n0 + i * 16 + 4 * fq;
          f4_t v = acc[i][j] * row_scale(j) + b4[i];
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
      // fp32 residual read-modify-write.  The residual reads of TWO 16-row blocks (8 x 16 B per lane) are issued
      // together, unguarded (rows clamped to M-1), in front of the LDS transposes that consume them: one exposed HBM
      // round trip per 32 rows instead of one per 8 (the guarded load -> add -> store chain was latency-bound:
      // ~49 k cycles per tile, more than a third of the O projection's time).
#pragma unroll
      for (int jb = 0; jb < RM; jb += 2) {
        f4_t hv[2][2][2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              int m = m0 + (jb + jj) * 16 + rd_row + 8 * t;
              m = m < a.M ? m : a.M - 1;
              hv[jj][hf][t] = *(const f4_t*)(a.resid + (size_t)m * a.N + n0 + hf * 32 + rd_ch * 4);
            }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = jb + jj;
          const float sj = row_scale(j);
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
            for (int il = 0; il < 2; ++il) {
              *(f4_t*)(scr + frow * SCR_STR + il * 64 + fq * 16) = acc[hf * 2 + il][j] * sj + b4[hf * 2 + il];
              acc[hf * 2 + il][j] = f4_t{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const f4_t v = *(const f4_t*)(scr + (rd_row + 8 * t) * SCR_STR + rd_ch * 16);
              const int m = m0 + j * 16 + rd_row + 8 * t;
              if (m < a.M) *(f4_t*)(a.resid + (size_t)m * a.N + n0 + hf * 32 + rd_ch * 4) = hv[jj][hf][t] + v;
            }
          }
        }
      }
    } else {
      // STORE (fused QKV): q keeps an fp16 lo plane (attention.hip rescales and re-splits it), k gets the c8 byte pairs
      // of the fp8-corrected QK^T (columns lo_c8_from .. lo_c8_to), v an fp16 lo plane again (attention's Vl·P pass).
      // GELU (FC1 -> FC2 operand): the lo plane is the c8 byte pair.
      [[maybe_unused]] const gelu_coef_t gk = gelu_coefficients();
      // one specialised copy of the loop per lo format of the tile (compile-time LOFMT: -1 = no lo plane), picked by a
      // wave-uniform branch per tile instead of a format test per element group
      // tiled output (OT): element offset = tile base (wave-uniform: row block tm, 64-column chunk of this wave, the
      // wave's 128-row half) + j·16 rows + this lane's (row, swizzled 16-byte chunk).  Rows rd_row and rd_row + 8 of a
      // 16-row group differ in bit 2 of the swizzle term ((row >> 1) & 7), i.e. in 32 halves of the chunk offset.
      constexpr bool TILED_OUT = OT;
      [[maybe_unused]] size_t tl_base = 0;
      [[maybe_unused]] unsigned tl_off[2] = {0, 0};
      if constexpr (TILED_OUT) {
        tl_base = (((size_t)tm * (a.N >> 6) + (n0 >> 6)) * 256 + wm * TM) * 64;
        tl_off[0] = (unsigned)(rd_row * 64 + ((rd_ch ^ (rd_row >> 1)) << 3));
        tl_off[1] = (tl_off[0] ^ 32u) + 8 * 64;
      }
      auto store_tile = [&](auto fmt_c) __attribute__((always_inline)) {
        constexpr int LOFMT = decltype(fmt_c)::value;
#pragma unroll
        for (int j = 0; j < RM; ++j) {
          [[maybe_unused]] h4_t lo4[RN];
          const float sj = row_scale(j);
#pragma unroll
          for (int i = 0; i < RN; ++i) {
            f4_t v = acc[i][j] * sj + b4[i];
            acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};
            if constexpr (EPI == ZK_EPI_GELU) {
              const gelu_f2_t g01 = gelu_erf2(gelu_f2_t{v[0], v[1]}, gk), g23 = gelu_erf2(gelu_f2_t{v[2], v[3]}, gk);
              v[0] = g01[0]; v[1] = g01[1]; v[2] = g23[0]; v[3] = g23[1];
            }
            h4_t hi;
            if constexpr (LOFMT >= 0) zk_pin(v);      // hi and lo must come from the same rounded value
#pragma unroll
            for (int e = 0; e < 4; ++e) hi[e] = (half_t)v[e];
            if constexpr (LOFMT >= 0) {
              const float vv[4] = {v[0], v[1], v[2], v[3]};
              lo4[i] = zk_lo4(vv, hi, LOFMT);
            }
            *(h4_t*)(scr + frow * SCR_STR + i * 32 + fq * 8) = hi;
          }
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const h8_t v = *(const h8_t*)(scr + (rd_row + 8 * t) * SCR_STR + rd_ch * 16);
            const int m = m0 + j * 16 + rd_row + 8 * t;
            // this lane's 8 halves in a plane: row-major, or the tile form (uniform tile base + 32-bit lane offset: see tl_off)
            const size_t oo = TILED_OUT ? tl_base + (size_t)(tl_off[t] + j * 1024) : (size_t)m * a.ldo + n0 + rd_ch * 8;
            // (non-temporal: the planes are read by the NEXT kernel from their first row on, long after these lines would have
            // left the caches; kept out of L2 they stop evicting the W / X lines of the k-loops: FC1 +2 %, QKV +0.7 %)
#ifdef ZK_C8_NO_NT      // probe builds: plain stores
            if (m < a.M) *(h8_t*)(a.o_hi + oo) = v;
#else
            if (m < a.M) __builtin_nontemporal_store(v, (h8_t*)(a.o_hi + oo));
#endif
          }
          if constexpr (LOFMT >= 0) {
#pragma unroll
            for (int i = 0; i < RN; ++i) *(h4_t*)(scr + frow * SCR_STR + i * 32 + fq * 8) = lo4[i];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const h8_t v = *(const h8_t*)(scr + (rd_row + 8 * t) * SCR_STR + rd_ch * 16);
              const int m = m0 + j * 16 + rd_row + 8 * t;
              const size_t oo = TILED_OUT ? tl_base + (size_t)(tl_off[t] + j * 1024) : (size_t)m * a.ldo + n0 + rd_ch * 8;
#ifdef ZK_C8_NO_NT
              if (m < a.M) *(h8_t*)(a.o_lo + oo) = v;
#else
              if (m < a.M) __builtin_nontemporal_store(v, (h8_t*)(a.o_lo + oo));
#endif
            }
          }
        }
      };
      const bool want_lo = a.o_lo != nullptr && n0 < a.lo_n_limit;
      if (!want_lo) store_tile(std::integral_constant<int, -1>{});
      else if (EPI == ZK_EPI_GELU || (n0 >= a.lo_c8_from && n0 < a.lo_c8_to)) store_tile(std::integral_constant<int, ZK_LO_C8>{});
      else if constexpr (EPI != ZK_EPI_GELU) store_tile(std::integral_constant<int, ZK_LO_F16>{});
    }
    ++c_ord; c_tile += stride;
  };


  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using H0 = K0;
  using H1 = K1;

  // prologue: step 0 (main, slot 0) lands before anything is read
#if ZK_C8_ROLES
  const bool is_loader = ZK_C8_RLOAD ? wave >= 4 : wave < 4;
#ifdef ZK_C8_RPRIO      // probe: static priority for one role (1 = the piece-free waves, 2 = the loaders)
  if ((ZK_C8_RPRIO == 2) == is_loader) __builtin_amdgcn_s_setprio(1);
#endif
  if (is_loader) {
#pragma unroll
    for (int p = 0; p < 16; ++p) issue_piece_r(K0{}, p);
  }
#else
#pragma unroll
  for (int pc = 0; pc < LPT; ++pc) issue_piece(K0{}, pc);
#endif
  advance_load(K0{});
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  int c_k = 0;
  bool epi_pending = false;

  // one ring step of kind KIND (its slot), prefetching the following step (kind KIND^1) into the other slot
  // ROLE: 0 = every wave issues its 8 pieces (production), 1 = loader role (16 pieces), 2 = no pieces
  auto step = [&](auto kind_c, auto role_c) __attribute__((always_inline)) {
    constexpr int KIND = decltype(kind_c)::value;
    constexpr int ROLE = decltype(role_c)::value;
    using KN = std::integral_constant<int, KIND ^ 1>;
    // the pieces to issue in front of MFMA group g (of G) of chunk c
    auto pieces = [&](auto c_c, auto g_c, auto G_c) __attribute__((always_inline)) {
      constexpr int c = decltype(c_c)::value, g = decltype(g_c)::value, G = decltype(G_c)::value;
      if constexpr (ROLE == 0) {
        if constexpr (c8_piece(c, g, G) >= 0) issue_piece(KN{}, c8_piece(c, g, G));
      }
#if ZK_C8_ROLES
      else if constexpr (ROLE == 1) {
        constexpr int n = c8r_cnt(c);
        static_for<n>([&](auto qc) __attribute__((always_inline)) {
          constexpr int q = decltype(qc)::value;
          if constexpr (q * G / n == g) issue_piece_r(KN{}, c8r_base(c) + q);
        });
      }
#endif
    };
    using CM1 = std::integral_constant<int, -1>;
    using GRN = std::integral_constant<int, RN>;
    using G2RN = std::integral_constant<int, 2 * RN>;
    // X tile 0 of this step, then per W tile i: the deferred MFMA of the previous step (old wf[i], X tile RM-1 in
    // xf[1]) followed by the read of this step's W fragment INTO wf[i] — one W register set serves both steps
    ld_frag(xf[0], kind_c, xad, std::integral_constant<int, 0>{});
    static_for<RN>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      pieces(CM1{}, ic, GRN{});
      mma(KN{}, H0{}, acc[i][RM - 1], wf[i], xf[1]);      // (very first step: zero fragments)
      mma(KN{}, H1{}, acc[i][RM - 1], wf[i], xf[1]);
      ld_frag(wf[i], kind_c, wad, ic);
    });
    if constexpr (KIND == 0) {
      if (epi_pending) {
        epilogue();
        epi_pending = false;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // chunks j = 0 .. RM-2: issue the reads of X tile j+1, wait for everything older (tile j; in chunk 0 also the W
    // fragments), then the MFMAs of tile j with the LDS-DMA pieces of the next step in between (c8_piece)
    static_for<RM - 1>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      ld_frag(xf[(j + 1) & 1], kind_c, xad, std::integral_constant<int, j + 1>{});
      if constexpr (j == 0)
        asm volatile("s_waitcnt lgkmcnt(2)"
                     : "+v"(xf[0].a), "+v"(xf[0].b), "+v"(wf[0].a), "+v"(wf[0].b), "+v"(wf[1].a), "+v"(wf[1].b),
                       "+v"(wf[2].a), "+v"(wf[2].b), "+v"(wf[3].a), "+v"(wf[3].b));
      else
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(xf[j & 1].a), "+v"(xf[j & 1].b));
      __builtin_amdgcn_sched_barrier(0);
      static_for<RN>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        if constexpr (KIND == 1) pieces(jc, ic, GRN{}); else pieces(jc, ic, G2RN{});
        mma(kind_c, H0{}, acc[i][j], wf[i], xf[j & 1]);
      });
      if constexpr (KIND == 0) {
        static_for<RN>([&](auto ic) __attribute__((always_inline)) {
          constexpr int i = decltype(ic)::value;
          pieces(jc, std::integral_constant<int, RN + i>{}, G2RN{});
          mma(kind_c, H1{}, acc[i][j], wf[i], xf[j & 1]);
        });
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    advance_load(KN{});
    if (KIND == 1) {
      if (++c_k == nk) { c_k = 0; epi_pending = true; }
    }
    // the next step must have landed; X tile RM-1 of this step (xf[1]) and wf[] are in registers
#if defined(ZK_C8_STAMPS) && ZK_C8_STAMPS == 1      // (ZK_C8_STAMPS=2: only the kernel-level clock stamps, the steps run as shipped)
    unsigned long long st_a, st_b, st_c;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_a) : : "memory");
    wait_vmcnt<0>();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_b) : : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xf[1].a), "+v"(xf[1].b) : : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_c) : : "memory");
    {
      unsigned* sp = zk_c8_stamp_buf + (size_t)((blockIdx.x * 8 + wave) * 64 + (l_step & 63)) * 4;
      const unsigned a0 = (unsigned)st_a, b0 = (unsigned)st_b, c0 = (unsigned)st_c, id = (unsigned)l_step * 2 + KIND;
      asm volatile("s_store_dword %1, %0, 0x0\n\ts_store_dword %2, %0, 0x4\n\ts_store_dword %3, %0, 0x8\n\ts_store_dword %4, %0, 0xc"
                   : : "s"(sp), "s"(a0), "s"(b0), "s"(c0), "s"(id) : "memory");
    }
#else
    wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xf[1].a), "+v"(xf[1].b) : : "memory");      // X tile RM-1 is in xf[1]
    __builtin_amdgcn_s_barrier();
#endif
  };

#pragma unroll
  for (int i = 0; i < RN; ++i) wf[i].a = wf[i].b = i4v_t{0, 0, 0, 0};
  xf[1].a = xf[1].b = i4v_t{0, 0, 0, 0};
#if ZK_C8_ROLES
  if (is_loader) {
    for (int c_step = 0; c_step < total; c_step += 2) {
      step(K0{}, std::integral_constant<int, 1>{});
      step(K1{}, std::integral_constant<int, 1>{});
    }
  } else {
    for (int c_step = 0; c_step < total; c_step += 2) {
      step(K0{}, std::integral_constant<int, 2>{});
      step(K1{}, std::integral_constant<int, 2>{});
    }
  }
#else
  for (int c_step = 0; c_step < total; c_step += 2) {
    step(K0{}, K0{});
    step(K1{}, K0{});
  }
#endif
  // deferred tile of the last step (kind 1)
#pragma unroll
  for (int i = 0; i < RN; ++i) mma(K1{}, H0{}, acc[i][RM - 1], wf[i], xf[1]);
  epilogue();
#ifdef ZK_C8_STAMPS
  {
    unsigned long long t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) : : "memory");
    unsigned long long* cp = zk_c8_clock_buf + blockIdx.x * 4;
    asm volatile("s_store_dwordx2 %1, %0, 0x10\n\ts_store_dwordx2 %2, %0, 0x18" : : "s"(cp), "s"(t1), "s"(r1) : "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
#endif
}

template <int EPI, bool XT = false, bool OT = false>
void launch_cfg(const zk_gemm_args& a, hipStream_t s) {
  constexpr int lds = 2 * (256 + 256) * 128 + 2 * 8 * 256 + 8 * 16 * 144 + 8 * 512;
  auto k = gemm_c8_kernel<EPI, XT, OT>;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
  const int ntiles = ((a.M + 255) / 256) * (a.N / 256);
  const int grid = ntiles < 256 ? ((ntiles + 7) / 8) * 8 : 256;
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, s, a);
}

}  // namespace

// Host-side shape contract as zk_launch_gemm: N % 256 == 0, K % 64 == 0, M >= 1; x_lo / w_lo are c8 planes.  The kernel
// stages the last 256-row block of x WHOLE (no M-tail clamp in the loader), so the planes must be allocated for
// ceil(M/256)*256 rows: the caller states the allocation in x_rows and the launch is refused if it falls short.
int zk_launch_gemm_c8(const zk_gemm_args& a_in, int epi, hipStream_t s) {
  zk_gemm_args a = a_in;
  if (a.ldo == 0) a.ldo = a.N;
  if (a.M < 1 || a.N % 256 || a.K % 64 || a.ldo < a.N || a.ldo % 8) return -1;
  if (a.x_rows < (int64_t)((a.M + 255) / 256) * 256) return -1;
  switch (epi) {
    case ZK_EPI_STORE: if (a.x_tiled) launch_cfg<ZK_EPI_STORE, true>(a, s); else launch_cfg<ZK_EPI_STORE>(a, s); break;
    case ZK_EPI_GELU:
      if (a.x_tiled && a.o_tiled) launch_cfg<ZK_EPI_GELU, true, true>(a, s);
      else if (a.o_tiled) launch_cfg<ZK_EPI_GELU, false, true>(a, s);
      else if (a.x_tiled) launch_cfg<ZK_EPI_GELU, true, false>(a, s);
      else launch_cfg<ZK_EPI_GELU>(a, s);
      break;
    case ZK_EPI_RESID: if (a.x_tiled) launch_cfg<ZK_EPI_RESID, true>(a, s); else launch_cfg<ZK_EPI_RESID>(a, s); break;
    default: launch_cfg<ZK_EPI_PATCH>(a, s); break;
  }
  return 0;
}
