// pss_xcorr_i8.hip -- the PSS correlation for RTL-SDR (u8 I/Q) capture buffers on the int8 matrix cores.
//
// An RTL-SDR sample is the integer (u8 - 127) scaled by 1/128.  The fp32 template taps of one template are
// scaled by a per-template constant q and rounded to 24-bit integers (|T_int| <= 8.3e6: quantisation error
// <= 6e-8 of the template's largest tap, below fp32 rounding), which split EXACTLY into three signed base-256
// digits.  v_mfma_i32_16x16x64_i8 multiplies int8 samples with int8 digits and accumulates in int32 -- exact
// integer arithmetic, no summation-order effects.  The three digit sums are recombined as
// ((S2 * 256) + S1) * 256 + S0 in fp32 (two roundings), scaled by 1 / (128 q), squared and accumulated like
// the other kernels.  16x16x64 consumes 32 taps per instruction at ~2x the bf16 rate: 240 MFMAs per
// wave-window instead of 432 for the bf16 three-term kernel.
//
// Operands.  Real GEMM with K = 2 * taps.  The capture buffer is stored as int8 pairs a = 127 - u8 (so that
// all 256 codes fit: -128 .. 127; the correlation changes sign, its power does not) in natural (re, im) order =
// the A operand; B_re = digits of (tr, -ti), B_im = digits of (ti, tr), so one A operand feeds both
// accumulators.  Lane (i, kg) of an A operand holds the 8 consecutive samples lag_i + 32 kb + 8 kg .. +7
// (16 bytes); the Toeplitz operand of (lag sub-tile mt, tap block kb) depends on mt + 2 kb only.  Samples
// are 2 bytes, lanes start at any sample: LDS holds the window twice, the second copy shifted by one sample,
// so that every lane reads 4 aligned dwords from the copy matching its parity.
//
// Tiling: 256-thread workgroup = 512 output positions x one 16-template group, 8
// sub-tiles per wave; per window 3 digit passes x 5 tap blocks, fully unrolled; the window's B operands
// (30 KB) sit in LDS next to the capture samples.  Digit 2 has its own int32 accumulator; digits 1
// and 0 share one (shifted left by 8 between the passes), so int -> float conversion happens twice per
// output and window, not per digit.
//
// k_xcorr_i8x3: B operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4), one window ahead; the
// operands of tap block e + 1 are read from LDS behind the first MFMA pair of block e; the digit passes walk
// the tap blocks boustrophedon.
#include "lcs_internal.h"
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define NW LCS_NW_MAX
#define NFM LCS_NF_MAX
#define GM LCS_G_MAX

#ifndef I8_MT
#define I8_MT 8                                          // 16-lag sub-tiles per wave
#endif
#define I8_LAGS (4 * I8_MT * 16)
#define I8_TILES ((LCS_N_IDX + I8_LAGS - 1) / I8_LAGS)
#define I8_NKB LCS_I8_KB                                  // 32-tap blocks per window: 137 taps + spread <= 160
#define I8_AW (I8_LAGS + 32 * I8_NKB + 32)                // staged samples per window
#define I8_ADW (((I8_AW / 2 + 63) / 64) * 64)             // dwords per staged copy: whole 64-dword LDS-DMA chunks
// I8_ADW is a multiple of 32, + 16 puts the shifted copy 16 banks away from the natural one, so the even-lag lanes
// (natural copy, banks 0..12 of a 32-lane group) and the odd-lag lanes (shifted copy) of one ds_read never meet
// on a bank (with + 2 they did: SQ_LDS_BANK_CONFLICT = 16 % of the kernel's cycles)
#define I8_ACOPY (I8_ADW + 16)
#define I8_QMAX 8300000.0                                 // |T_int| bound: three balanced base-256 digits reach 8 355 711

// Per template (slot, foi, t): q = I8_QMAX / max tap magnitude; sc = 1 / (128 q) converts the integer
// correlation back to the reference's units.
__global__ __launch_bounds__(256) void k_i8_scales(const float2 *__restrict__ tmpl, double *__restrict__ tq,
                                                   float *__restrict__ sc, XcGeom geo) {
  LCS_TAIL_PRIO();
  const int slot = blockIdx.x;
  __shared__ float part[LCS_G_MAX * LCS_TG][4];
  for (int e = threadIdx.x; e < geo.G * LCS_TG * 4; e += 256) {      // 4 threads per template, 35 taps each
    const int col = e >> 2, qd = e & 3;            // column index: group col / 16, column col % 16
    const int c = lcs_col_tmpl(geo, col >> 4, col & 15);
    float mx = 0.f;
    if (c >= 0) {
      const int foi = c / 3, t = c % 3;
      const float2 *T = tmpl + (((size_t)slot * NFM + foi) * 3 + t) * 137;
      for (int m = qd * 35; m < min(137, qd * 35 + 35); ++m) mx = fmaxf(mx, fmaxf(fabsf(T[m].x), fabsf(T[m].y)));
    }
    part[col][qd] = mx;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < geo.G * LCS_TG; c += 256) {
    const float mx = fmaxf(fmaxf(part[c][0], part[c][1]), fmaxf(part[c][2], part[c][3]));
    const double q = (mx > 0.f) ? I8_QMAX / (double)mx : 0.0;
    tq[(size_t)slot * GM * LCS_TG + c] = q;
    sc[(size_t)slot * GM * LCS_TG + c] = (q > 0.0) ? (float)(1.0 / (128.0 * q)) : 0.f;
  }
}

__device__ __forceinline__ void digits3(int v, int &d0, int &d1, int &d2) {   // v = d0 + 256 d1 + 65536 d2, digits in [-128, 127]
  d0 = ((v + 128) & 255) - 128;
  const int v1 = (v - d0) >> 8;          // exact: v - d0 is a multiple of 256
  d1 = ((v1 + 128) & 255) - 128;
  d2 = (v1 - d1) >> 8;
}

// bt8[slot][w][g][digit][kb][op][lane] (uint4 = 16 int8): lane (n, kg) holds taps 32 kb + 8 kg .. +7 of template
// column c = 16 g + n delayed by start[w][foi(c)] - smin[w][g] (zero outside its 137 taps), as digit `digit` of
// the integer pairs (tr, -ti) (op 0, real output) or (ti, tr) (op 1, imaginary output).
__global__ __launch_bounds__(256) void k_fill_btab_i8(const float2 *__restrict__ tmpl, const int *__restrict__ start,
                                                      const int *__restrict__ smin, const double *__restrict__ tq,
                                                      uint4 *__restrict__ bt8, XcGeom geo) {
  LCS_TAIL_PRIO();
  const int slot = blockIdx.z;
  const int wg = blockIdx.y;
  const int w = wg / geo.G, g = wg % geo.G;
  const int s0 = smin[((size_t)slot * NW + w) * GM + g];
  uint4 *out = bt8 + (((size_t)slot * geo.n_comb + w) * geo.G + g) * (size_t)(3 * I8_NKB * 2 * 64);
  // one thread per (tap block, lane, half of the lane's 8 taps): 4 taps = 8 operand bytes per digit and output.
  // 30 VGPRs: under the 48 that two resident correlation workgroups leave free on a SIMD (512 - 2 x 232), so these
  // workgroups start beside them instead of waiting for one to retire; the table is stored with non-temporal
  // stores (it is read ~1.5 ms later by another kernel; allocating 173 MB of it in L2 only evicts the running
  // correlation's operands).  Together -1 % step time in the pipelined chain -- what this kernel costs there is its
  // memory traffic (skip-kernel ablation: 45 us per batch before, ~35 after).
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < I8_NKB * 64 * 2; e += gridDim.x * blockDim.x) {
    const int hf = e & 1, lane = (e >> 1) & 63, kb = e >> 7;
    const int c = lcs_col_tmpl(geo, g, lane & 15), kg = lane >> 4;
    int tr[4], ti[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) { tr[m] = 0; ti[m] = 0; }
    if (c >= 0) {
      const int foi = c / 3, t = c % 3;
      const int delta = start[((size_t)slot * NW + w) * NFM + foi] - s0;
      const double q = tq[(size_t)slot * GM * LCS_TG + g * LCS_TG + (lane & 15)];      // scales are stored per column
      const float2 *T = tmpl + (((size_t)slot * NFM + foi) * 3 + t) * 137;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int tap = 32 * kb + 8 * kg + 4 * hf + m - delta;
        if (tap >= 0 && tap < 137) { tr[m] = (int)rint((double)T[tap].x * q); ti[m] = (int)rint((double)T[tap].y * q); }
      }
    }
#pragma unroll
    for (int op = 0; op < 2; ++op) {                  // op 0: pairs (tr, -ti), op 1: pairs (ti, tr)
      uint32_t pk[3][2];
#pragma unroll
      for (int d = 0; d < 3; ++d) { pk[d][0] = 0u; pk[d][1] = 0u; }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int m = j >> 1;
        const int v = op ? ((j & 1) ? tr[m] : ti[m]) : ((j & 1) ? -ti[m] : tr[m]);
        int d0, d1, d2;
        digits3(v, d0, d1, d2);
        const int sh = 8 * (j & 3);
        pk[0][j >> 2] |= (uint32_t)(d0 & 255) << sh;
        pk[1][j >> 2] |= (uint32_t)(d1 & 255) << sh;
        pk[2][j >> 2] |= (uint32_t)(d2 & 255) << sh;
      }
#pragma unroll
      for (int d = 0; d < 3; ++d)
      {
        typedef unsigned u2v __attribute__((ext_vector_type(2)));
        u2v val = {pk[d][0], pk[d][1]};
        __builtin_nontemporal_store(val, reinterpret_cast<u2v *>(out + (((size_t)d * I8_NKB + kb) * 2 + op) * 64 + lane) + hf);
      }
    }
  }
}

// Everything a window needs reaches LDS by LDS-DMA (global_load_lds: no staging registers, no ds_write pass), issued
// one window ahead right behind the barrier: the B operands (30 KB, dwordx4 chunks of the per-window table) and the
// capture samples -- two copies of the window, natural and shifted by one sample, each a run of dwords (= sample pairs)
// taken from cap8 or cap8s, whichever holds the window start dword aligned.  The registers that frees pay for a
// one-block-deep operand prefetch: the B operands and the two new A operands of tap block e + 1 are read from LDS
// behind the first MFMA pair of block e, so the LDS latency sits under 14 MFMAs instead of in front of every
// block.  The digit passes walk the tap blocks boustrophedon (digit 2: kb 0..4, digit 1: kb 4..0, digit 0: kb 0..4)
// so the sliding A window never restarts: 32 A-operand reads per window instead of 48.
// Epilogue work is spread under the MFMA stream where its inputs allow: the digit-2 sums are converted to float
// while the digit-1 pass runs, the << 8 of the shared digit-1/0 accumulator sits in front of each sub-tile's first
// digit-0 MFMA; what is left behind the last block is 6 VALU operations per output.
// Measured (isolated, 64 buffers, 16x16x64 issues every ~18 cycles: tools/microbench/mfma_rate.hip): the 15 tap
// blocks of a window run at the MFMA issue rate (0.059 ms per block-launch); the rest is per-window and
// per-workgroup cost (barrier, epilogue, prologue of each of the 15 workgroup rounds, last round 25 % full).
__global__ __launch_bounds__(256, 2) void k_xcorr_i8x3(const uint16_t *__restrict__ cap8, const uint16_t *__restrict__ cap8s,
                                                          const int *__restrict__ smin, const uint4 *__restrict__ bt8,
                                                          const float *__restrict__ sc, float *__restrict__ sg, XcGeom geo,
                                                          int slot0, int n_slots, int xcd_map) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int per_slot = I8_TILES * geo.G;
  int q, sidx;
  if (xcd_map) { sidx = blockIdx.x & 7; q = blockIdx.x >> 3; sidx += 8 * (q / per_slot); q = q % per_slot; }
  else { sidx = blockIdx.x / per_slot; q = blockIdx.x % per_slot; }
  if (sidx >= n_slots) return;
  const int slot = slot0 + sidx, g = q / I8_TILES, idx0 = (q % I8_TILES) * I8_LAGS;
  const int widx0 = idx0 + wave * (I8_MT * 16);

  __shared__ uint32_t ldsA[2][2][I8_ACOPY];
  constexpr int NBLK = 3 * I8_NKB;
  constexpr int BW = NBLK * 2 * 64;       // uint4 per window (30 KB), table order [digit][kb][op][lane]
  constexpr int NCH = BW / 64;            // 1 KiB chunks: one global_load_lds_dwordx4 per wave each
  constexpr int NCA = 2 * (I8_ADW / 64);  // 256-byte chunks of the two sample copies: one global_load_lds_dword per wave each
  __shared__ uint4 ldsB[2][BW];
  const size_t cstride = lcs_cap8_stride(geo.n_cap);
  const uint32_t *capd = reinterpret_cast<const uint32_t *>(cap8 + (size_t)slot * cstride) + lane;     // dword j = samples (2j, 2j+1)
  const uint32_t *capsd = reinterpret_cast<const uint32_t *>(cap8s + (size_t)slot * cstride) + lane;   // dword j = samples (2j+1, 2j+2)
  const int *smin_s = smin + (size_t)slot * NW * GM + g;
  const uint4 *bt_s = bt8 + ((size_t)slot * geo.n_comb * geo.G + g) * (size_t)BW + lane;
  const size_t bt_wstride = (size_t)geo.G * BW;
  const float my_sc = sc[(size_t)slot * GM * LCS_TG + g * LCS_TG + (lane & 15)];
  const int p0 = wave * (I8_MT * 16) + (lane & 15) + 8 * (lane >> 4);
  const int par = p0 & 1;
  const int a_dw = (p0 - par) >> 1;

  f32x4 P[I8_MT];
#pragma unroll
  for (int mt = 0; mt < I8_MT; ++mt) P[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // window W: samples L0 .. of this slot as the natural copy (dword i = samples L0 + 2i, L0 + 2i + 1) and the shifted
  // one (L0 + 2i + 1, L0 + 2i + 2); an odd L0 swaps the roles of cap8 and cap8s
#define I8_DMA(W)                                                                                            \
  {                                                                                                          \
    const int L0_ = idx0 + smin_s[(W) * GM], h_ = L0_ >> 1;                                                  \
    const uint32_t *nat_ = ((L0_ & 1) ? capsd : capd) + h_;                                                  \
    const uint32_t *shf_ = (L0_ & 1) ? capd + h_ + 1 : capsd + h_;                                           \
    _Pragma("unroll") for (int c_ = 0; c_ < (NCA + 3) / 4; ++c_) {                                           \
      const int ca_ = wave + 4 * c_;          /* chunk = (copy, 64-dword piece) */                           \
      if (ca_ < NCA) {                                                                                       \
        const int cp_ = ca_ / (I8_ADW / 64), k_ = ca_ % (I8_ADW / 64);                                       \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((cp_ ? shf_ : nat_) + 64 * k_), \
                                         (__attribute__((address_space(3))) void *)(ldsA[(W) & 1][cp_] + 64 * k_), 4, 0, 0); \
      }                                                                                                      \
    }                                                                                                        \
    uint4 *dst_ = ldsB[(W) & 1];                                                                             \
    const uint4 *src_ = bt_s + (size_t)(W) * bt_wstride;                                                     \
    _Pragma("unroll") for (int c_ = 0; c_ < (NCH + 3) / 4; ++c_) {                                           \
      const int ch_ = wave + 4 * c_;          /* chunk = (digit, kb, op) in table order */                   \
      if (ch_ < NCH)                                                                                         \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src_ + ch_ * 64),  \
                                         (__attribute__((address_space(3))) void *)(dst_ + ch_ * 64), 16, 0, 0); \
    }                                                                                                        \
  }
#define I8_RD_A(S) { const uint32_t *p_ = bufA + 8 * (S); Aw[S] = (i32x4){(int)p_[0], (int)p_[1], (int)p_[2], (int)p_[3]}; }
#define I8_RD_B(E, Q)                                                                                        \
  {                                                                                                          \
    constexpr int d_ = (E) / I8_NKB, j_ = (E) % I8_NKB, kb_ = (d_ == 1) ? I8_NKB - 1 - j_ : j_;              \
    constexpr int tb_ = (2 - d_) * I8_NKB + kb_;                                                             \
    _Pragma("unroll") for (int op_ = 0; op_ < 2; ++op_) {                                                    \
      const uint4 t_ = bl[(tb_ * 2 + op_) * 64];                                                             \
      Bq[Q][op_] = (i32x4){(int)t_.x, (int)t_.y, (int)t_.z, (int)t_.w};                                      \
    }                                                                                                        \
  }
  I8_DMA(0);
  for (int w = 0; w < geo.n_comb; ++w) {
    __syncthreads();                       // drains this wave's LDS-DMA chunks of window w (vmcnt(0)), then everybody's
    if (w + 1 < geo.n_comb) I8_DMA(w + 1); // buffers (w + 1) & 1: last read in window w - 1
    const uint4 *bl = ldsB[w & 1] + lane;
    const uint32_t *bufA = ldsA[w & 1][par] + a_dw;
    // digit 2 accumulates into (tR, tI); digits 1 and 0 share one int32 accumulator: after the digit-1 pass it is
    // shifted left by 8 and the digit-0 products are added on top (|S1| <= 274 * 128 * 128 = 4.5e6, so
    // 256 S1 + S0 stays below 2^31): one int -> float conversion per digit group instead of per digit.
    i32x4 tR[I8_MT], tI[I8_MT], aR[I8_MT], aI[I8_MT];
    f32x4 fR[I8_MT], fI[I8_MT];            // float(S2): exact, |S2| < 2^24
    i32x4 Aw[2 * I8_NKB + I8_MT - 2];
    i32x4 Bq[2][2];
#pragma unroll
    for (int s = 0; s < I8_MT; ++s) I8_RD_A(s);
    I8_RD_B(0, 0);
#define I8_PF_MFMA(MT)                                                                                       \
  {                                                                                                          \
    if (d == 0) {                                                                                            \
      const i32x4 cr = (j == 0) ? (i32x4){0, 0, 0, 0} : tR[MT];                                              \
      const i32x4 ci = (j == 0) ? (i32x4){0, 0, 0, 0} : tI[MT];                                              \
      tR[MT] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Aw[2 * kb + (MT)], Bq[e & 1][0], cr, 0, 0, 0);          \
      tI[MT] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Aw[2 * kb + (MT)], Bq[e & 1][1], ci, 0, 0, 0);          \
    } else {                                                                                                 \
      const i32x4 cr = (d == 1 && j == 0) ? (i32x4){0, 0, 0, 0} : (d == 2 && j == 0) ? aR[MT] << 8 : aR[MT]; \
      const i32x4 ci = (d == 1 && j == 0) ? (i32x4){0, 0, 0, 0} : (d == 2 && j == 0) ? aI[MT] << 8 : aI[MT]; \
      aR[MT] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Aw[2 * kb + (MT)], Bq[e & 1][0], cr, 0, 0, 0);          \
      aI[MT] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Aw[2 * kb + (MT)], Bq[e & 1][1], ci, 0, 0, 0);          \
    }                                                                                                        \
  }
#pragma unroll
    for (int e = 0; e < NBLK; ++e) {
      const int d = e / I8_NKB, j = e % I8_NKB;
      const int kb = (d == 1) ? I8_NKB - 1 - j : j;
      // The first MFMA pair of block e carries the wait for block e's operands (read during block e - 1); the reads
      // for block e + 1 are issued behind it, so that every s_waitcnt lgkmcnt(0) the compiler places finds only
      // reads that have had 14 MFMAs to complete.
      I8_PF_MFMA(0);
      __builtin_amdgcn_sched_barrier(0);
      if (e + 1 < NBLK) {                  // operands of block e + 1
        const int d1 = (e + 1) / I8_NKB, j1 = (e + 1) % I8_NKB;
        const int kb1 = (d1 == 1) ? I8_NKB - 1 - j1 : j1;
        {
          const int tb = (2 - d1) * I8_NKB + kb1;
#pragma unroll
          for (int op = 0; op < 2; ++op) {
            const uint4 t_ = bl[(tb * 2 + op) * 64];
            Bq[(e + 1) & 1][op] = (i32x4){(int)t_.x, (int)t_.y, (int)t_.z, (int)t_.w};
          }
        }
        if (kb1 > kb) { I8_RD_A(2 * kb1 + I8_MT - 2); I8_RD_A(2 * kb1 + I8_MT - 1); }
        else if (kb1 < kb) { I8_RD_A(2 * kb1); I8_RD_A(2 * kb1 + 1); }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 1; mt < I8_MT; ++mt) I8_PF_MFMA(mt);
      if (d == 1) {                        // digit-1 pass: the finished digit-2 sums of sub-tiles j, j + NKB go to float
#pragma unroll
        for (int mt = j; mt < I8_MT; mt += I8_NKB)
#pragma unroll
          for (int r = 0; r < 4; ++r) { fR[mt][r] = (float)tR[mt][r]; fI[mt][r] = (float)tI[mt][r]; }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef I8_PF_MFMA
#pragma unroll
    for (int mt = 0; mt < I8_MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {     // integer correlation S2 * 65536 + (256 S1 + S0) in fp32; scaled by 1 / (128 q)^2 at the end
        const float xr = fmaf(fR[mt][r], 65536.f, (float)aR[mt][r]);
        const float xi = fmaf(fI[mt][r], 65536.f, (float)aI[mt][r]);
        P[mt][r] = fmaf(xi, xi, fmaf(xr, xr, P[mt][r]));
      }
  }
#undef I8_DMA
#undef I8_RD_A
#undef I8_RD_B
  const float ncomb = (float)geo.n_comb;
  float *o = sg + (((size_t)slot * geo.G + g) * LCS_N_IDX) * LCS_TG + (lane & 15);
#pragma unroll
  for (int mt = 0; mt < I8_MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int idx = widx0 + mt * 16 + 4 * (lane >> 4) + r;
      if (idx < LCS_N_IDX) o[(size_t)idx * LCS_TG] = __fdiv_rn(P[mt][r] * (my_sc * my_sc), ncomb);
    }
}

int lcs_launch_fill_btab_i8(lcs_ctx *c, int n_buf, const XcGeom &geo) {
  hipLaunchKernelGGL(k_i8_scales, dim3(n_buf), dim3(256), 0, c->stream, c->tmpl, c->tq, c->tsc, geo);
  hipLaunchKernelGGL(k_fill_btab_i8, dim3((I8_NKB * 128 + 255) / 256, geo.n_comb * geo.G, n_buf), dim3(256), 0, c->stream, c->tmpl, c->start, c->smin,
                     c->tq, c->bt8, geo);
  HIPCHK(c, hipGetLastError());
  return LCS_OK;
}
int lcs_launch_xcorr_i8(lcs_ctx *c, hipStream_t sxc, const XcGeom &geo, int slot0, int n_slots, int xcd_map) {
  const unsigned grid = (unsigned)(I8_TILES * geo.G * n_slots);
  hipLaunchKernelGGL(k_xcorr_i8x3, dim3(grid), dim3(256), 0, sxc, c->cap8, c->cap8s, c->smin, c->bt8, c->tsc, c->single, geo, slot0,
                     n_slots, xcd_map);
  HIPCHK(c, hipGetLastError());
  // executed work: per wave and window 3 digits x I8_NKB tap blocks x I8_MT sub-tiles x (re, im) MFMAs of 16x16x64 MACs
  c->last_xc_ops += (double)grid * 4 * geo.n_comb * (3.0 * I8_NKB * I8_MT * 2) * (2.0 * 16 * 16 * 64);
  c->last_xc_kernel = "k_xcorr_i8x3";
  return LCS_OK;
}
