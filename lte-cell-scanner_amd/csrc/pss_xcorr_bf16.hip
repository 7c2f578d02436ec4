// pss_xcorr_bf16.hip -- the PSS correlation for capture buffers that are EXACT in bf16.
//
// RTL-SDR samples are (u8 - 127) / 128: eight significant bits, i.e. exactly representable in
// bfloat16.  The fp32 templates are split exactly into three bf16 terms (t = t1 + t2 + t3, 8 + 8 + 8
// significand bits), every product sample * t_i is exact in fp32, and the matrix cores accumulate in
// fp32: three v_mfma_f32_16x16x32_bf16 evaluate the same dot product as the fp32 kernel (same inputs,
// fp32 accumulation, different summation order: deviation ~1e-7 relative, parity bar 1e-5) at 16/3
// times the fp32 MFMA rate.  Same fused formulation as pss_xcorr.hip (correlation + 15-window
// incoherent combining, window-start delays folded into the template table).
//
// Operands.  The complex product is taken as a real GEMM with K = 2 * taps:
//   A[lag][2m], A[lag][2m+1] = xr[lag+m], xi[lag+m]      -- the capture buffer as it lies in memory
//   B_re[2m], B_re[2m+1] = tr[m], -ti[m] ;  B_im[2m], B_im[2m+1] = ti[m], tr[m]
// so one A operand feeds both the real and the imaginary accumulator.  A 16x16x32 MFMA consumes 16
// taps; lane (i, kg) of the A operand holds the four consecutive samples lag_i + 16 kb + 4 kg .. +3
// (16 bytes).  A is Toeplitz, so the operand of (lag sub-tile mt, tap block kb) depends on mt + kb
// only: a wave keeps a sliding window of 8 operands in registers and reads ONE new operand per tap
// block from LDS (4 dwords) for its 8 sub-tiles x 6 MFMAs.
//
// Tiling.  A 256-thread workgroup owns 512 output positions x one 16-template group; each wave 128
// positions (8 sub-tiles of 16).  Per window the capture samples are staged once into LDS (one barrier
// per window); the template operands (6 x 1 KB per tap block: {re, im} x 3 split terms) are read by
// every wave straight from L2/L1, one or two blocks ahead.
#include "lcs_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define NW LCS_NW_MAX
#define NFM LCS_NF_MAX
#define GM LCS_G_MAX

#define BF_LAGS (4 * BF_MT * 16)
#define BF_TILES ((LCS_N_IDX + BF_LAGS - 1) / BF_LAGS)
#ifndef BF_MT
#define BF_MT 8                                        // 16-lag sub-tiles per wave
#endif
#ifndef BF_WPS
#define BF_WPS 2                                       // waves per SIMD the unrolled kernel is compiled for
#endif
#define BF_AW (BF_LAGS + 16 * LCS_BF_KB_MAX + 16)     // staged samples per window
#define BF_OPS 6                                      // B operands per tap block: re1 re2 re3 im1 im2 im3
#define BF_TOPS 3                                     // stored per tap block: the three split terms as (tr, ti) pairs

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define AS_BF(x) __builtin_bit_cast(bf16x8, (x))

__device__ __forceinline__ uint32_t bf16_rne(float v) {   // finite inputs only
  const uint32_t u = __float_as_uint(v);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_val(uint32_t h) { return __uint_as_float(h << 16); }

// bt16[slot][w][g][kb][term][lane] (uint4 = 8 bf16): lane (n, kg) holds taps 16 kb + 4 kg .. +3 of template
// column c = 16 g + n delayed by start[w][foi(c)] - smin[w][g] (zero outside its 137 taps) as (tr, ti)
// pairs, one uint4 per split term.  The correlation kernels derive the two MFMA operands from it in
// registers: real output (tr, -ti) = sign flip of the high halves, imaginary output (ti, tr) = half swap.
__global__ __launch_bounds__(256) void k_fill_btab_bf16(const float2 *__restrict__ tmpl, const int *__restrict__ start,
                                                        const int *__restrict__ smin, const int *__restrict__ kp2,
                                                        uint4 *__restrict__ bt16, XcGeom geo) {
  LCS_TAIL_PRIO();
  const int slot = blockIdx.z;
  const int wg = blockIdx.y;
  const int w = wg / geo.G, g = wg % geo.G;
  const int k2 = kp2[((size_t)slot * NW + w) * GM + g];
  const int s0 = smin[((size_t)slot * NW + w) * GM + g];
  const int nkb = min((2 * k2 + 15) / 16, LCS_BF_KB_MAX);
  uint4 *out = bt16 + (((size_t)slot * geo.n_comb + w) * geo.G + g) * (size_t)(LCS_BF_KB_MAX * BF_TOPS * 64);
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nkb * 64; e += gridDim.x * blockDim.x) {
    const int kb = e >> 6, lane = e & 63;
    const int c = g * LCS_TG + (lane & 15), kg = lane >> 4;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (c < geo.n_tmpl) {
      const int foi = c / 3, t = c % 3;
      const int delta = start[((size_t)slot * NW + w) * NFM + foi] - s0;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int tap = 16 * kb + 4 * kg + m - delta;
        if (tap >= 0 && tap < 137) {
          const float2 T = tmpl[(((size_t)slot * NFM + foi) * 3 + t) * 137 + tap];
          v[2 * m] = T.x; v[2 * m + 1] = T.y;
        }
      }
    }
    uint32_t h[BF_TOPS][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float r = v[j];
#pragma unroll
      for (int sidx = 0; sidx < BF_TOPS; ++sidx) {          // exact three-term split: r = h1 + h2 + h3
        const uint32_t hb = bf16_rne(r);
        h[sidx][j] = hb;
        r = r - bf16_val(hb);
      }
    }
#pragma unroll
    for (int op = 0; op < BF_TOPS; ++op) {
      uint4 q;
      q.x = h[op][0] | (h[op][1] << 16); q.y = h[op][2] | (h[op][3] << 16);
      q.z = h[op][4] | (h[op][5] << 16); q.w = h[op][6] | (h[op][7] << 16);
      out[((size_t)kb * BF_TOPS + op) * 64 + lane] = q;
    }
  }
}

__device__ __forceinline__ float pow2sum_bf(float re, float im) { return fmaf(re, re, im * im); }
// (tr, ti) pairs -> operand of the real output (tr, -ti) and of the imaginary output (ti, tr)
__device__ __forceinline__ u32x4 bf_op_re(u32x4 x) { return x ^ (u32x4){0x80000000u, 0x80000000u, 0x80000000u, 0x80000000u}; }
__device__ __forceinline__ u32x4 bf_op_im(u32x4 x) {
  return (u32x4){__builtin_amdgcn_alignbit(x[0], x[0], 16), __builtin_amdgcn_alignbit(x[1], x[1], 16),
                 __builtin_amdgcn_alignbit(x[2], x[2], 16), __builtin_amdgcn_alignbit(x[3], x[3], 16)};
}

// General kernel (any tap-block count per window, a loop over the blocks): every wave reads its six B
// operands per tap block straight from global memory (the four waves of a workgroup hit the same
// lines within a microsecond: one L2 fetch, L1 hits for the rest), two blocks ahead.  No barrier
// inside a window, so the waves of a workgroup drift apart and keep the matrix pipe fed while one of
// them waits.  (Staging the operands through LDS with a barrier per block was 3 % slower.)
__global__ __launch_bounds__(256, 2) void k_xcorr_bf16x3_loop(const uint32_t *__restrict__ capb, const int *__restrict__ smin,
                                                                const int *__restrict__ kp2, const uint4 *__restrict__ bt16,
                                                                float *__restrict__ sg, XcGeom geo, int slot0, int n_slots,
                                                                int xcd_map) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per_slot = BF_TILES * geo.G;
  int q, sidx;
  if (xcd_map) { sidx = blockIdx.x & 7; q = blockIdx.x >> 3; sidx += 8 * (q / per_slot); q = q % per_slot; }
  else { sidx = blockIdx.x / per_slot; q = blockIdx.x % per_slot; }
  if (sidx >= n_slots) return;
  const int slot = slot0 + sidx, g = q / BF_TILES, idx0 = (q % BF_TILES) * BF_LAGS;
  const int widx0 = idx0 + wave * (BF_MT * 16);
  const int n_mt = min(max((LCS_N_IDX - widx0 + 15) / 16, 0), BF_MT);

  __shared__ uint32_t ldsA[2][BF_AW];
  const uint32_t *cap = capb + (size_t)slot * geo.n_cap;
  const int *smin_s = smin + (size_t)slot * NW * GM + g;
  const int *kp2_s = kp2 + (size_t)slot * NW * GM + g;
  const uint4 *bt_s = bt16 + ((size_t)slot * geo.n_comb * geo.G + g) * (size_t)(LCS_BF_KB_MAX * BF_TOPS * 64) + lane;
  const size_t bt_wstride = (size_t)geo.G * (LCS_BF_KB_MAX * BF_TOPS * 64);
  const int a_off = wave * (BF_MT * 16) + (lane & 15) + 4 * (lane >> 4);

  f32x4 P[BF_MT];
#pragma unroll
  for (int mt = 0; mt < BF_MT; ++mt) P[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int ASTEPS = (BF_AW + 255) / 256;
  uint32_t preA[ASTEPS];
#define BF_LOAD_A(W)                                                                   \
  {                                                                                    \
    const int L0_ = idx0 + smin_s[(W) * GM];                                           \
    _Pragma("unroll") for (int r_ = 0; r_ < ASTEPS; ++r_) {                            \
      const int n_ = tid + 256 * r_;                                                   \
      const uint32_t s_ = (uint32_t)(L0_ + n_);                                        \
      preA[r_] = (n_ < BF_AW && s_ < geo.n_cap) ? cap[s_] : 0u;                        \
    }                                                                                  \
  }
#define BF_GLOAD_B(DST, W, KB)                                                         \
  {                                                                                    \
    const uint4 *src_ = bt_s + (size_t)(W) * bt_wstride + (size_t)(KB) * (BF_TOPS * 64); \
    _Pragma("unroll") for (int op_ = 0; op_ < BF_TOPS; ++op_) {                        \
      const uint4 t_ = src_[op_ * 64];                                                 \
      (DST)[op_] = (u32x4){t_.x, t_.y, t_.z, t_.w};                                    \
    }                                                                                  \
  }
#define BF_READ_A(DST, S)                                                              \
  {                                                                                    \
    const uint32_t *p_ = bufA + a_off + 16 * (S);                                      \
    (DST) = (u32x4){p_[0], p_[1], p_[2], p_[3]};                                       \
  }
  // the sequence of tap blocks over all windows is walked with a two-deep register queue
  u32x4 Bq0[BF_TOPS], Bq1[BF_TOPS], Bnew[BF_TOPS];
  int nkb = min((2 * kp2_s[0] + 15) / 16, LCS_BF_KB_MAX);
  BF_LOAD_A(0);
  BF_GLOAD_B(Bq0, 0, 0);
  // position of the block that Bq1 / Bnew refer to
  int pw = 0, pk = 1, pn = nkb;                   // (window, block, blocks in that window) of the next block to request
  if (pk >= pn) { pw = 1; pk = 0; pn = (pw < geo.n_comb) ? min((2 * kp2_s[pw * GM] + 15) / 16, LCS_BF_KB_MAX) : 0; }
  if (pw < geo.n_comb) BF_GLOAD_B(Bq1, pw, pk);
  ++pk;
  if (pk >= pn) { ++pw; pk = 0; pn = (pw < geo.n_comb) ? min((2 * kp2_s[pw * GM] + 15) / 16, LCS_BF_KB_MAX) : 0; }
  for (int w = 0; w < geo.n_comb; ++w) {
    uint32_t *bufA = ldsA[w & 1];
#pragma unroll
    for (int r = 0; r < ASTEPS; ++r) {
      const int n = tid + 256 * r;
      if (n < BF_AW) bufA[n] = preA[r];
    }
    if (w + 1 < geo.n_comb) BF_LOAD_A(w + 1);
    __syncthreads();
    f32x4 aR[BF_MT], aI[BF_MT];
#pragma unroll
    for (int mt = 0; mt < BF_MT; ++mt) { aR[mt] = (f32x4){0.f, 0.f, 0.f, 0.f}; aI[mt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    u32x4 Aw[BF_MT];
#pragma unroll
    for (int s = 0; s < BF_MT - 1; ++s) BF_READ_A(Aw[s], s);
#pragma unroll 1
    for (int kb = 0; kb < nkb; ++kb) {
      if (pw < geo.n_comb) BF_GLOAD_B(Bnew, pw, pk);
      ++pk;
      if (pk >= pn) { ++pw; pk = 0; pn = (pw < geo.n_comb) ? min((2 * kp2_s[pw * GM] + 15) / 16, LCS_BF_KB_MAX) : 0; }
      BF_READ_A(Aw[BF_MT - 1], kb + BF_MT - 1);
      u32x4 Bop[BF_OPS];
#pragma unroll
      for (int t3 = 0; t3 < BF_TOPS; ++t3) { Bop[t3] = bf_op_re(Bq0[t3]); Bop[3 + t3] = bf_op_im(Bq0[t3]); }
#pragma unroll
      for (int mt = 0; mt < BF_MT; ++mt) {
        if (mt < n_mt) {
          aR[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AS_BF(Aw[mt]), AS_BF(Bop[0]), aR[mt], 0, 0, 0);
          aI[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AS_BF(Aw[mt]), AS_BF(Bop[3]), aI[mt], 0, 0, 0);
          aR[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AS_BF(Aw[mt]), AS_BF(Bop[1]), aR[mt], 0, 0, 0);
          aI[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AS_BF(Aw[mt]), AS_BF(Bop[4]), aI[mt], 0, 0, 0);
          aR[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AS_BF(Aw[mt]), AS_BF(Bop[2]), aR[mt], 0, 0, 0);
          aI[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AS_BF(Aw[mt]), AS_BF(Bop[5]), aI[mt], 0, 0, 0);
        }
      }
#pragma unroll
      for (int s = 0; s < BF_MT - 1; ++s) Aw[s] = Aw[s + 1];
#pragma unroll
      for (int op = 0; op < BF_TOPS; ++op) { Bq0[op] = Bq1[op]; Bq1[op] = Bnew[op]; }
    }
#pragma unroll
    for (int mt = 0; mt < BF_MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) P[mt][r] = P[mt][r] + pow2sum_bf(aR[mt][r], aI[mt][r]);
    nkb = (w + 1 < geo.n_comb) ? min((2 * kp2_s[(w + 1) * GM] + 15) / 16, LCS_BF_KB_MAX) : 0;
  }
#undef BF_LOAD_A
#undef BF_GLOAD_B
#undef BF_READ_A
  const float ncomb = (float)geo.n_comb;
  float *o = sg + (((size_t)slot * geo.G + g) * LCS_N_IDX) * LCS_TG + (lane & 15);
#pragma unroll
  for (int mt = 0; mt < BF_MT; ++mt) {
    if (mt < n_mt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int idx = widx0 + mt * 16 + 4 * (lane >> 4) + r;
        if (idx < LCS_N_IDX) o[(size_t)idx * LCS_TG] = __fdiv_rn(P[mt][r], ncomb);
      }
    }
  }
}

// Straight-line kernel for the common case that every (window, group) has the same number of tap
// blocks (9 when 137 taps + window-start spread <= 144: any +-ppm grid of the CLI): the blocks of a
// window are unrolled, so the Toeplitz window and the operand queue are pure register renaming (no
// v_mov rotation), and the 48 MFMAs of a block are ordered split-term-major so that MFMAs into the
// same accumulator are 16 instructions apart.  1.78 ms per 64-buffer launch against 1.95 ms for the loop.
// (sub-tiles past idx 9599 in the last workgroup of a row are computed and dropped: no branch in the block)
#define BF_MFMA_BLOCK(AW, S0, BT, FIRST)                                                                  \
  _Pragma("unroll") for (int sp_ = 0; sp_ < 3; ++sp_) {                                                   \
    const u32x4 bre_ = bf_op_re(BT[sp_]), bim_ = bf_op_im(BT[sp_]);                                       \
    _Pragma("unroll") for (int mt_ = 0; mt_ < BF_MT; ++mt_) {                                             \
      /* the first MFMA of a window starts from the inline constant 0 instead of a zeroed register */     \
      const f32x4 cr_ = ((FIRST) && sp_ == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : aR[mt_];                    \
      const f32x4 ci_ = ((FIRST) && sp_ == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : aI[mt_];                    \
      aR[mt_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AS_BF(AW[(S0) + mt_]), AS_BF(bre_), cr_, 0, 0, 0); \
      aI[mt_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AS_BF(AW[(S0) + mt_]), AS_BF(bim_), ci_, 0, 0, 0); \
    }                                                                                                     \
  }
template <int NKB, int NWV>   // NKB tap blocks per window, the same for every (window, group) of the launch (host-checked); NWV waves
__global__ __launch_bounds__(NWV * 64, BF_WPS) void k_xcorr_bf16x3_unrolled(const uint32_t *__restrict__ capb, const int *__restrict__ smin,
                                                                  const int *__restrict__ kp2, const uint4 *__restrict__ bt16,
                                                                  float *__restrict__ sg, XcGeom geo, int slot0, int n_slots,
                                                                  int xcd_map) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NT = NWV * 64, LAGS = NWV * BF_MT * 16, TILES = (LCS_N_IDX + LAGS - 1) / LAGS, AWN = LAGS + 16 * LCS_BF_KB_MAX + 16;
  const int per_slot = TILES * geo.G;
  int q, sidx;
  if (xcd_map) { sidx = blockIdx.x & 7; q = blockIdx.x >> 3; sidx += 8 * (q / per_slot); q = q % per_slot; }
  else { sidx = blockIdx.x / per_slot; q = blockIdx.x % per_slot; }
  if (sidx >= n_slots) return;
  const int slot = slot0 + sidx, g = q / TILES, idx0 = (q % TILES) * LAGS;
  const int widx0 = idx0 + wave * (BF_MT * 16);
  const int n_mt = min(max((LCS_N_IDX - widx0 + 15) / 16, 0), BF_MT);

  __shared__ uint32_t ldsA[2][AWN];
  const uint32_t *cap = capb + (size_t)slot * geo.n_cap;
  const int *smin_s = smin + (size_t)slot * NW * GM + g;
  (void)kp2;
  const uint4 *bt_s = bt16 + ((size_t)slot * geo.n_comb * geo.G + g) * (size_t)(LCS_BF_KB_MAX * BF_TOPS * 64) + lane;
  const size_t bt_wstride = (size_t)geo.G * (LCS_BF_KB_MAX * BF_TOPS * 64);
  const int a_off = wave * (BF_MT * 16) + (lane & 15) + 4 * (lane >> 4);

  f32x4 P[BF_MT];
#pragma unroll
  for (int mt = 0; mt < BF_MT; ++mt) P[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int ASTEPS = (AWN + NT - 1) / NT;
  uint32_t preA[ASTEPS];
#define BF_LOAD_A(W)                                                                   \
  {                                                                                    \
    const int L0_ = idx0 + smin_s[(W) * GM];                                           \
    _Pragma("unroll") for (int r_ = 0; r_ < ASTEPS; ++r_) {                            \
      const int n_ = tid + NT * r_;                                                    \
      const uint32_t s_ = (uint32_t)(L0_ + n_);                                        \
      preA[r_] = (n_ < AWN && s_ < geo.n_cap) ? cap[s_] : 0u;                          \
    }                                                                                  \
  }
#define BF_GLOAD_B(DST, W, KB)                                                         \
  {                                                                                    \
    const uint4 *src_ = bt_s + (size_t)(W) * bt_wstride + (size_t)(KB) * (BF_TOPS * 64); \
    _Pragma("unroll") for (int op_ = 0; op_ < BF_TOPS; ++op_) {                        \
      const uint4 t_ = src_[op_ * 64];                                                 \
      (DST)[op_] = (u32x4){t_.x, t_.y, t_.z, t_.w};                                    \
    }                                                                                  \
  }
#ifndef LCS_BF16_DEPTH
#define LCS_BF16_DEPTH 2          // operand blocks requested ahead of use
#endif
  constexpr int PD = LCS_BF16_DEPTH;
  u32x4 Bq[NKB + PD][BF_TOPS];    // static indices only: block kb of the window, +PD = first ones of the next
  BF_LOAD_A(0);
#pragma unroll
  for (int i = 0; i < PD; ++i) BF_GLOAD_B(Bq[i], 0, i);
  for (int w = 0; w < geo.n_comb; ++w) {
    const bool has_next = w + 1 < geo.n_comb;
    uint32_t *bufA = ldsA[w & 1];
#pragma unroll
    for (int r = 0; r < ASTEPS; ++r) {
      const int n = tid + NT * r;
      if (n < AWN) bufA[n] = preA[r];
    }
    if (has_next) BF_LOAD_A(w + 1);
    __syncthreads();
    f32x4 aR[BF_MT], aI[BF_MT];
    u32x4 Aw[NKB + BF_MT - 1];
#pragma unroll
    for (int s = 0; s < BF_MT - 1; ++s) { const uint32_t *p_ = bufA + a_off + 16 * s; Aw[s] = (u32x4){p_[0], p_[1], p_[2], p_[3]}; }
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      // request the block PD ahead: in this window, or one of the first PD blocks of the next one
      if (kb + PD < NKB) BF_GLOAD_B(Bq[kb + PD], w, kb + PD)
      else if (has_next) BF_GLOAD_B(Bq[kb + PD], w + 1, kb + PD - NKB)
      { const uint32_t *p_ = bufA + a_off + 16 * (kb + BF_MT - 1); Aw[kb + BF_MT - 1] = (u32x4){p_[0], p_[1], p_[2], p_[3]}; }
      BF_MFMA_BLOCK(Aw, kb, Bq[kb], kb == 0);
#ifndef LCS_BF16_NO_SCHED_BARRIER
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
#pragma unroll
    for (int op = 0; op < BF_TOPS; ++op)
#pragma unroll
      for (int i = 0; i < PD; ++i) Bq[i][op] = Bq[NKB + i][op];
#pragma unroll
    for (int mt = 0; mt < BF_MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) P[mt][r] = P[mt][r] + pow2sum_bf(aR[mt][r], aI[mt][r]);
  }
#undef BF_LOAD_A
#undef BF_GLOAD_B
  const float ncomb = (float)geo.n_comb;
  float *o = sg + (((size_t)slot * geo.G + g) * LCS_N_IDX) * LCS_TG + (lane & 15);
#pragma unroll
  for (int mt = 0; mt < BF_MT; ++mt) {
    if (mt < n_mt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int idx = widx0 + mt * 16 + 4 * (lane >> 4) + r;
        if (idx < LCS_N_IDX) o[(size_t)idx * LCS_TG] = __fdiv_rn(P[mt][r], ncomb);
      }
    }
  }
}

// Launch both kernels of the bf16 path for slots [0, n_buf) on `sxc`; table fill goes to the main stream.
int lcs_launch_fill_btab_bf16(lcs_ctx *c, int n_buf, const XcGeom &geo) {
  hipLaunchKernelGGL(k_fill_btab_bf16, dim3(4, geo.n_comb * geo.G, n_buf), dim3(256), 0, c->stream, c->tmpl, c->start, c->smin,
                     c->kp2, c->bt16, geo);
  HIPCHK(c, hipGetLastError());
  return LCS_OK;
}
int lcs_launch_xcorr_bf16(lcs_ctx *c, hipStream_t sxc, const XcGeom &geo, int slot0, int n_slots, int xcd_map) {
  const unsigned grid = (unsigned)(BF_TILES * geo.G * n_slots);
  static const bool force_loop = getenv("LCS_BF16_LOOP") != nullptr;   // tuning knob
  // every (window, group) has exactly 9 tap blocks (137 taps + spread <= 144): the straight-line kernel, 4 waves per
  // workgroup (1- and 2-wave workgroups run the kernel as fast in isolation but leave the small kernels of the
  // neighbouring batches fewer openings: -13 % in the pipelined chain)
  if (c->grid_max_k2 <= 72 && !force_loop)
    hipLaunchKernelGGL((k_xcorr_bf16x3_unrolled<9, 4>), dim3(grid), dim3(256), 0, sxc, c->capb, c->smin, c->kp2, c->bt16, c->single,
                       geo, slot0, n_slots, xcd_map);
  else
    hipLaunchKernelGGL(k_xcorr_bf16x3_loop, dim3(grid), dim3(256), 0, sxc, c->capb, c->smin, c->kp2, c->bt16, c->single, geo,
                       slot0, n_slots, xcd_map);
  HIPCHK(c, hipGetLastError());
  return LCS_OK;
}
