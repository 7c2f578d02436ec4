// pss_xcorr_f16.hip -- the PSS correlation for capture buffers that are NOT dongle bytes (complex<float> sources already
// resident in HBM), on the fp16 matrix cores at fp32 accuracy.
//
// A float sample has 24 significant bits, an fp16 operand 11.  Scaled by a power of two per buffer (so that the largest
// component lies in [512, 1024): exact), every sample component x splits into xh = fp16(x) and xl = fp16(x - xh): xh + xl
// carries 22 bits (relative error 2^-22 = 2.4e-7, the two roundings of an fp32 product chain).  The templates split the
// same way (per-template power-of-two scale).  Then  x t = xh th + xh tl + xl th + xl tl  with the last term below 2^-22 of
// the first: three v_mfma_f32_16x16x32_f16 per operand pair, every product exact in the fp32 accumulator, summed in
// fp32 by the matrix pipe.  That is 3 x the dense fp16 rate (2.5 PFLOP/s) against the fp32 MFMA rate (157 TFLOP/s) of
// k_xcorr_mfma_blk: ~5 x less matrix-pipe time for the same 1e-6 agreement with the reference.
//
// Formulation as pss_xcorr_i8.hip: real GEMM with K = 2 taps, the capture buffer in natural (re, im) order = the A operand
// (one fp16 pair = one dword per sample, so a window starts at any sample without the shifted copies the 2-byte int8
// pairs need), B_re = (tr, -ti), B_im = (ti, tr), the window-start spread of a template group folded into the B table as
// delays.  v_mfma_f32_16x16x32_f16 takes K = 32 = 16 taps: lane (i, kg) of an A operand holds samples lag_i + 16 kb +
// 4 kg .. + 3 (16 bytes, read from LDS at 4-byte alignment), the operand of (sub-tile mt, tap block kb) depends on mt + kb
// only and slides by one operand per block.  256-thread workgroup = 512 lags x one 16-template group, 8 sub-tiles per
// wave; a window is 10 tap blocks (160 taps).  The B operands are resident rows (below: one 46 KB image per (buffer,
// group), loaded once per workgroup), the samples are staged per wave one window ahead; 64 KB of LDS, 64 fp32
// accumulators + 32 power sums per lane: no digit recombination, the epilogue is two FMAs per output and window.
#include "lcs_internal.h"
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define NW LCS_NW_MAX
#define NFM geo.n_f      // strides of the per-hypothesis / per-group tables: the call's own grid
#define GM geo.G

#define F16_MT 8
#define F16_LAGS (4 * F16_MT * 16)
#define F16_TILES ((LCS_N_IDX + F16_LAGS - 1) / F16_LAGS)
#define F16_NKB 10                                        // 16-tap blocks per window: 137 taps + spread <= 160
#define F16_TARGET_EXP 9                                  // operands are scaled into [2^9, 2^10)

__device__ __forceinline__ uint32_t f16_bits(_Float16 h) { return (uint32_t)__builtin_bit_cast(unsigned short, h); }
__device__ __forceinline__ void f16_split(float v, _Float16 &hi, _Float16 &lo) {
  hi = (_Float16)v;
  lo = (_Float16)(v - (float)hi);
}

// ---- per buffer: the power of two that brings the largest sample component into [512, 1024).
// Round 4: no copy at all when the caller's buffers can be read in place (even n_cap, 16-byte aligned): the fp64 stages read
// them through CapSrc (lcs_cap_src), so only the per-buffer maximum is taken here -- a read-only pass with four 16-byte loads
// in flight per lane.
#define F16_MAX_ILP 4           // independent 16-byte loads per lane, all in flight before the first use
#define F16_MAXP 128            // partial maxima per buffer (workgroups of k_f16_max4 per buffer: 75 for 153600 samples)
__global__ __launch_bounds__(256) void k_f16_max4(const float4 *__restrict__ src, uint32_t n_cap, unsigned *__restrict__ xpart) {
  LCS_TAIL_PRIO();
  __shared__ float wmax[4];
  const int slot = blockIdx.y;
  const uint32_t n4 = n_cap / 2;
  const float4 *b = src + (size_t)slot * n4;
  // workgroup w covers float4s [w * 1024, w * 1024 + 1024): lane t takes t, t + 256, t + 512, t + 768 (no loop, four
  // loads outstanding per lane, 64 KB per wave-instruction group coalesced)
  const uint32_t i0 = blockIdx.x * (256 * F16_MAX_ILP) + threadIdx.x;
  float4 v[F16_MAX_ILP];
#pragma unroll
  for (int k = 0; k < F16_MAX_ILP; ++k) {
    const uint32_t i = i0 + 256 * k;
    v[k] = b[min(i, n4 - 1)];                 // unconditional (a branch per load would serialise them); the tail re-reads the last pair
  }
  float m = 0.f;
#pragma unroll
  for (int k = 0; k < F16_MAX_ILP; ++k) m = fmaxf(m, fmaxf(fmaxf(fabsf(v[k].x), fabsf(v[k].y)), fmaxf(fabsf(v[k].z), fabsf(v[k].w))));
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off));
  // one plain store per workgroup: 300 atomicMax per buffer on ONE address took 145 us per 64 buffers, 7 x the pass itself
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float w = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    xpart[(size_t)slot * F16_MAXP + blockIdx.x] = (w > 0.f && w < INFINITY) ? __float_as_uint(w) : 0u;     // non-negative floats order like their bits
  }
}
__global__ __launch_bounds__(128) void k_f16_max_fold(const unsigned *__restrict__ xpart, int n_part, unsigned *__restrict__ xmax_bits) {
  LCS_TAIL_PRIO();
  const int slot = blockIdx.x, t = threadIdx.x;
  __shared__ unsigned w2[2];
  unsigned m = (t < n_part) ? xpart[(size_t)slot * F16_MAXP + t] : 0u;
  for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_down((int)m, off));
  if ((t & 63) == 0) w2[t >> 6] = m;
  __syncthreads();
  if (t == 0) xmax_bits[slot] = max(w2[0], w2[1]);
}
// the same maximum over an existing cap32 (odd n_cap)
__global__ __launch_bounds__(256) void k_f16_max(const float2 *__restrict__ cap32, uint32_t n_cap, unsigned *__restrict__ xmax_bits) {
  LCS_TAIL_PRIO();
  const int slot = blockIdx.y;
  const float2 *c = cap32 + (size_t)slot * n_cap;
  float m = 0.f;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_cap; i += gridDim.x * blockDim.x) {
    const float2 v = c[i];
    m = fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y)));
  }
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off));
  if ((threadIdx.x & 63) == 0 && m > 0.f && m < INFINITY) atomicMax(xmax_bits + slot, __float_as_uint(m));     // non-negative floats order like their bits
}
__device__ __forceinline__ int f16_scale_exp(float mx) { return (mx > 0.f) ? F16_TARGET_EXP - ilogbf(mx) : 0; }

// complex<float> buffers (the caller's, or cap32) -> (re, im) fp16 pairs, hi and lo parts, zero-padded behind n_cap like the
// int8 copies (the LDS-DMA of the correlation reads past the end).  PAIRS: one 16-byte load and two 8-byte stores per lane
// (even n_cap, 16-byte aligned source; the slot stride of the fp16 copies is a multiple of 8 samples).
__device__ __forceinline__ void f16_split_sample(float2 v, int k, uint32_t &h, uint32_t &l) {
  _Float16 rh, rl, ih, il;
  f16_split(ldexpf(v.x, k), rh, rl);
  f16_split(ldexpf(v.y, k), ih, il);
  h = f16_bits(rh) | (f16_bits(ih) << 16);
  l = f16_bits(rl) | (f16_bits(il) << 16);
}
template <bool PAIRS>
__global__ __launch_bounds__(256) void k_f16_ingest(const float2 *__restrict__ src32, uint32_t n_cap, const unsigned *__restrict__ xmax_bits,
                                                    uint32_t *__restrict__ cap16h, uint32_t *__restrict__ cap16l) {
  LCS_TAIL_PRIO();
  const int slot = blockIdx.y;
  const size_t stride = lcs_cap8_stride(n_cap);
  const int k = f16_scale_exp(__uint_as_float(xmax_bits[slot]));
  const float2 *c = src32 + (size_t)slot * n_cap;
  if (PAIRS) {
    // workgroup w covers sample pairs [w * 1024, w * 1024 + 1024) of the padded slot, four independent pairs per lane
    const float4 *c4 = reinterpret_cast<const float4 *>(c);
    uint2 *oh = reinterpret_cast<uint2 *>(cap16h + (size_t)slot * stride), *ol = reinterpret_cast<uint2 *>(cap16l + (size_t)slot * stride);
    const size_t i0 = (size_t)blockIdx.x * (256 * F16_MAX_ILP) + threadIdx.x;
    float4 v[F16_MAX_ILP];
#pragma unroll
    for (int q = 0; q < F16_MAX_ILP; ++q) {
      const size_t i = i0 + 256 * q;
      v[q] = c4[min(i, (size_t)n_cap / 2 - 1)];        // unconditional loads: all four in flight; lanes past the data mask below
    }
#pragma unroll
    for (int q = 0; q < F16_MAX_ILP; ++q) {
      const size_t i = i0 + 256 * q;
      if (i >= stride / 2) continue;
      uint2 h = make_uint2(0u, 0u), l = make_uint2(0u, 0u);
      if (2 * i < n_cap) {
        f16_split_sample(make_float2(v[q].x, v[q].y), k, h.x, l.x);
        f16_split_sample(make_float2(v[q].z, v[q].w), k, h.y, l.y);
      }
      oh[i] = h;
      ol[i] = l;
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < stride; i += (size_t)gridDim.x * blockDim.x) {
      uint32_t h = 0, l = 0;
      if (i < n_cap) f16_split_sample(c[i], k, h, l);
      cap16h[(size_t)slot * stride + i] = h;
      cap16l[(size_t)slot * stride + i] = l;
    }
  }
}

// per template column: power-of-two scale exponent of the template (largest tap into [512, 1024)) and the factor that
// takes the scaled correlation back to the reference's units, 2^-(k_x + k_t)
__global__ __launch_bounds__(256) void k_f16_scales(const float2 *__restrict__ tmpl, const unsigned *__restrict__ xmax_bits, int *__restrict__ texp,
                                                    float *__restrict__ sc, XcGeom geo) {
  LCS_TAIL_PRIO();
  const int slot = blockIdx.x;
  const int kx = f16_scale_exp(__uint_as_float(xmax_bits[slot]));
  for (int col = threadIdx.x; col < geo.G * LCS_TG; col += blockDim.x) {
    const int c = lcs_col_tmpl(geo, col >> 4, col & 15);
    float mx = 0.f;
    if (c >= 0) {
      const float2 *T = tmpl + (((size_t)slot * NFM + c / 3) * 3 + c % 3) * 137;
      for (int m = 0; m < 137; ++m) mx = fmaxf(mx, fmaxf(fabsf(T[m].x), fabsf(T[m].y)));
    }
    const int kt = f16_scale_exp(mx);
    texp[(size_t)slot * GM * LCS_TG + col] = kt;
    sc[(size_t)slot * GM * LCS_TG + col] = (c >= 0) ? ldexpf(1.f, -(kx + kt)) : 0.f;
  }
}

// Resident operand rows, as in pss_xcorr_i8.hip: across the windows of a buffer a template column changes only by its
// delay inside its group, so a workgroup loads ONE operand image per (buffer, group) into LDS at its start (46 KB) and
// every lane reads its 4 taps of tap block kb at dword 16 kb + 4 kg + F16R_OFF - delay of its column's row.  An fp16
// (re, im) pair is one dword, so any integer delay is dword aligned: one copy of every row (the int8 kernel's 2-byte
// pairs need two).  brow16[slot][g][term][op][row(n)][i]: position i = tap i - F16R_OFF; term 0 = fp16(value), term 1 =
// fp16(value - term 0); op 0: pairs (tr, -ti) (real output), op 1: pairs (ti, tr) (imaginary output).  Rows of
// consecutive columns sit 184 dwords apart plus n >> 2: 184 n = -8 n (mod 32) gives banks {0, 24, 16, 8} + (n >> 2), so
// the 32 lanes of a dword read (16 columns x 2 tap quads, 4 banks apart) fall on 32 different banks when their delays agree.
// (Rounds 3-4a: a 40 KB operand table per (window, group), copied into LDS in two chunks per window behind two barriers.)
#define F16R_OFF LCS_I8_OFF
#define F16R_ROW 184
#define F16R_RLEN 176
#define F16R_BLK 2948
#define F16R_IMG LCS_F16_IMG
static_assert(F16R_RLEN >= 16 * (F16_NKB - 1) + 12 + F16R_OFF + 4 && F16R_BLK >= 15 * F16R_ROW + 3 + F16R_RLEN && F16R_IMG >= 4 * F16R_BLK &&
              F16R_IMG % 64 == 0 && F16R_ROW % 32 == 24 && 137 + F16R_OFF - 1 <= 16 * F16_NKB, "operand image geometry");
__host__ __device__ static inline int f16r_rowoff(int n) { return n * F16R_ROW + (n >> 2); }

__global__ __launch_bounds__(256) void k_fill_brow_f16(const float2 *__restrict__ tmpl, const int *__restrict__ texp, uint32_t *__restrict__ brow,
                                                       XcGeom geo) {
  LCS_TAIL_PRIO();
  const int slot = blockIdx.z, g = blockIdx.y;
  uint32_t *out = brow + ((size_t)slot * geo.G + g) * F16R_IMG;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < LCS_TG * F16R_RLEN; e += gridDim.x * blockDim.x) {
    const int n = e / F16R_RLEN, i = e % F16R_RLEN;
    const int c = lcs_col_tmpl(geo, g, n), tap = i - F16R_OFF;
    float tr = 0.f, ti = 0.f;
    if (c >= 0 && tap >= 0 && tap < 137) {
      const int kt = texp[(size_t)slot * GM * LCS_TG + g * LCS_TG + n];
      const float2 T = tmpl[(((size_t)slot * NFM + c / 3) * 3 + c % 3) * 137 + tap];
      tr = ldexpf(T.x, kt); ti = ldexpf(T.y, kt);
    }
#pragma unroll
    for (int op = 0; op < 2; ++op) {
      const float a = op ? ti : tr, b = op ? tr : -ti;       // (tr, -ti) or (ti, tr)
      _Float16 ah, al, bh, bl;
      f16_split(a, ah, al);
      f16_split(b, bh, bl);
      out[(0 * 2 + op) * F16R_BLK + f16r_rowoff(n) + i] = f16_bits(ah) | (f16_bits(bh) << 16);
      out[(1 * 2 + op) * F16R_BLK + f16r_rowoff(n) + i] = f16_bits(al) | (f16_bits(bl) << 16);
    }
  }
}

// One barrier behind the image load; from there on every wave stages its OWN samples (hi and lo arrays of 128 lags + 160
// taps, two global_load_lds_dwordx4 of 36 lanes each) one window ahead into one of two LDS VARIABLES -- the window loop is
// unrolled by two so that the LDS-DMA destination and the array the operands are read from are different objects at
// compile time (with one array indexed by w & 1 the compiler waits for the next window's samples in front of the first
// operand read; the same happens when the operands are read as under-aligned 16-byte vectors instead of dwords, or when
// the LDS-DMA instructions of a window sit in more than one conditional block) -- and waits only for its own LDS-DMA.
// Per window 10 tap blocks x 3 products x 8 sub-tiles x (re, im) MFMAs.
#define F16_PA 288                                        // dwords (= samples) of a wave's staged array: 27 + 16 * 16 + 4 <= 288
static_assert(F16_PA >= 15 + 12 + 16 * (F16_NKB - 1 + F16_MT - 1) + 4 && F16_PA % 8 == 0 && F16_PA / 8 <= 64, "sample staging");
// NARROW: every window of the launch is narrow (geo.n_narrow == geo.n_comb: 137 taps + delay end below position 144): nine tap blocks, not ten.
template <bool NARROW>
__global__ __launch_bounds__(256, 2) void k_xcorr_f16x3(const uint32_t *__restrict__ cap16h, const uint32_t *__restrict__ cap16l,
                                                        const int *__restrict__ smin, const int *__restrict__ start,
                                                        const uint32_t *__restrict__ brow, const float *__restrict__ sc, float *__restrict__ sg,
                                                        XcGeom geo, int slot0, int n_slots, int xcd_map) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int per_slot = F16_TILES * geo.G;
  int q, sidx;
  if (xcd_map) { sidx = blockIdx.x & 7; q = blockIdx.x >> 3; sidx += 8 * (q / per_slot); q = q % per_slot; }
  else { sidx = blockIdx.x / per_slot; q = blockIdx.x % per_slot; }
  if (sidx >= n_slots) return;
  const int slot = slot0 + sidx, g = q / F16_TILES, idx0 = (q % F16_TILES) * F16_LAGS;
  const int widx0 = idx0 + wave * (F16_MT * 16);

  __shared__ uint32_t ldsA0[4][2][F16_PA], ldsA1[4][2][F16_PA];      // per wave: [hi, lo][sample], two buffers = two variables
  __shared__ uint32_t ldsR[F16R_IMG];
  const size_t cstride = lcs_cap8_stride(geo.n_cap);
  const uint32_t *caph = cap16h + (size_t)slot * cstride + 4 * lane, *capl = cap16l + (size_t)slot * cstride + 4 * lane;
  const int *smin_s = smin + (size_t)slot * NW * GM + g;
  const int col = lcs_col_tmpl(geo, g, lane & 15);
  const int *start_l = start + (size_t)slot * NW * NFM + (col >= 0 ? col / 3 : 0);
  const int rowbase = f16r_rowoff(lane & 15) + 4 * (lane >> 4) + F16R_OFF;
  {
    const uint32_t *img = brow + ((size_t)slot * geo.G + g) * F16R_IMG;
    constexpr int NCI = F16R_IMG / 256, NTL = (F16R_IMG - NCI * 256) / 64;
    static_assert(F16R_IMG == NCI * 256 + NTL * 64 && NTL < 4, "image = whole 1 KiB chunks + up to three 64-dword pieces");
#pragma unroll
    for (int c_ = 0; c_ < (NCI + 3) / 4; ++c_) {
      const int ch_ = wave + 4 * c_;
      if (ch_ < NCI)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(img + ch_ * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void *)(ldsR + ch_ * 256), 16, 0, 0);
    }
    if (wave < NTL)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(img + NCI * 256 + wave * 64 + lane),
                                       (__attribute__((address_space(3))) void *)(ldsR + NCI * 256 + wave * 64), 4, 0, 0);
  }
  int st_next = start_l[0];              // this lane's window start, fetched one window ahead
  const float my_sc = sc[(size_t)slot * GM * LCS_TG + g * LCS_TG + (lane & 15)];
  const int p0 = (lane & 15) + 4 * (lane >> 4);     // first sample of this lane's operand 0 in the wave's staged window

  f32x4 P[F16_MT];
#pragma unroll
  for (int mt = 0; mt < F16_MT; ++mt) P[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto dma = [&](int W, uint32_t (*dst)[2][F16_PA]) __attribute__((always_inline)) {
    const int L0_ = widx0 + smin_s[W * GM];
    if (lane < F16_PA / 8) {               // 36 lanes x 16 bytes = half an array per instruction
#pragma unroll
      for (int hl = 0; hl < 2; ++hl)
#pragma unroll
        for (int half = 0; half < 2; ++half)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((hl ? capl : caph) + (L0_ + half * (F16_PA / 2))),
                                           (__attribute__((address_space(3))) void *)(dst[wave][hl] + half * (F16_PA / 2)), 16, 0, 0);
    }
  };
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  // 16 bytes = 4 samples from a 4-byte aligned LDS address (the window starts at any sample), read as four dwords
#define F16_RD_AH(U) { const uint32_t *ph_ = bufAh + 16 * (U); const u32x4 th_ = {ph_[0], ph_[1], ph_[2], ph_[3]}; Ah[(U) % F16_MT] = __builtin_bit_cast(h8, th_); }
#define F16_RD_AL(U) { const uint32_t *pl_ = bufAl + 16 * (U); const u32x4 tl_ = {pl_[0], pl_[1], pl_[2], pl_[3]}; Al[(U) % F16_MT] = __builtin_bit_cast(h8, tl_); }
#define F16_RD_B(KB, TERM, DST)                                                                              \
    _Pragma("unroll") for (int op_ = 0; op_ < 2; ++op_)                                                      \
      { const uint32_t *q_ = bl + ((TERM) * 2 + op_) * F16R_BLK + 16 * (KB);                                  \
        const u32x4 t_ = {q_[0], q_[1], q_[2], q_[3]}; DST[op_] = __builtin_bit_cast(h8, t_); }

  auto window = [&](int w, const uint32_t (*rdA)[2][F16_PA], uint32_t (*wrA)[2][F16_PA]) __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);    // vmcnt(0), other counters untouched: this wave's own samples of window w have landed
    __builtin_amdgcn_sched_barrier(0);
    // nothing is in flight here: the delay fetched during the last window is consumed before anything new is requested
    const int dl = (col >= 0) ? st_next - smin_s[w * GM] : 0;
    const uint32_t *bl = ldsR + rowbase - dl;
    __builtin_amdgcn_sched_barrier(0);
    if (w + 1 < geo.n_comb) { st_next = start_l[(w + 1) * NFM]; dma(w + 1, wrA); }
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t *bufAh = rdA[wave][0] + p0, *bufAl = rdA[wave][1] + p0;
    f32x4 aR[F16_MT], aI[F16_MT];
    h8 Ah[F16_MT], Al[F16_MT];                // operands u = kb .. kb + 7 of the current block; operand kb + 8 replaces operand kb once its last product is issued
    // two operand sets rotate: block kb holds its hi term in S[kb & 1]; behind the first MFMA pair of product 0 the lo term
    // goes into the other set (product 1 multiplies it), behind the first pair of product 2 (hi term again) the NEXT block's
    // hi term replaces it: every B read has 14+ MFMAs to land, in 16 registers.  The samples slide the same way: the hi part
    // of operand kb is last used by the first pair of product 1, the lo part by the first pair of product 2.
    h8 S[2][2];                               // [set][op]
    constexpr int NKBW = NARROW ? F16_NKB - 1 : F16_NKB;
#pragma unroll
    for (int u = 0; u < F16_MT; ++u) { F16_RD_AH(u); F16_RD_AL(u); }
    F16_RD_B(0, 0, S[0]);
#pragma unroll
    for (int kb = 0; kb < NKBW; ++kb) {
      // three products per output: xh th, xh tl, xl th; sixteen independent accumulators between two uses of one
#pragma unroll
      for (int pr = 0; pr < 3; ++pr) {
#pragma unroll
        for (int mt = 0; mt < F16_MT; ++mt) {
          const h8 a = (pr == 2) ? Al[(kb + mt) % F16_MT] : Ah[(kb + mt) % F16_MT];
          const int term = (pr == 1) ? 1 : 0;
          const f32x4 cr = (kb == 0 && pr == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : aR[mt];
          const f32x4 ci = (kb == 0 && pr == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : aI[mt];
          const int set = (kb + term) & 1;
          aR[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, S[set][0], cr, 0, 0, 0);
          aI[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, S[set][1], ci, 0, 0, 0);
          if (mt == 0) {                      // behind a product's first MFMA pair: the next operands
            __builtin_amdgcn_sched_barrier(0);
            if (pr == 0) { F16_RD_B(kb, 1, S[(kb + 1) & 1]); }
            else if (kb + 1 < NKBW) {
              if (pr == 1) { F16_RD_AH(kb + F16_MT); }
              else { F16_RD_B(kb + 1, 0, S[(kb + 1) & 1]); F16_RD_AL(kb + F16_MT); }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
#pragma unroll
    for (int mt = 0; mt < F16_MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) P[mt][r] = fmaf(aI[mt][r], aI[mt][r], fmaf(aR[mt][r], aR[mt][r], P[mt][r]));
  };
  dma(0, ldsA0);
  __syncthreads();                         // the operand image is complete (every wave's chunks: vmcnt(0) in front of the barrier)
  if (widx0 >= LCS_N_IDX) return;          // the fourth wave of the 19th lag tile owns no lag; nothing below synchronises with it
  for (int w = 0; w < geo.n_comb; w += 2) {
    window(w, ldsA0, ldsA1);
    if (w + 1 < geo.n_comb) window(w + 1, ldsA1, ldsA0);
  }
#undef F16_RD_AH
#undef F16_RD_AL
#undef F16_RD_B
  const float ncomb = (float)geo.n_comb;
  float *o = sg + (((size_t)slot * geo.G + g) * LCS_N_IDX) * LCS_TG + (lane & 15);
#pragma unroll
  for (int mt = 0; mt < F16_MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int idx = widx0 + mt * 16 + 4 * (lane >> 4) + r;
      if (idx < LCS_N_IDX) o[(size_t)idx * LCS_TG] = __fdiv_rn((P[mt][r] * my_sc) * my_sc, ncomb);      // powers of two: exact, in two steps against underflow
    }
}

// ---- launchers -----------------------------------------------------------------------------------------------------
// d_src: the caller's complex<float> buffers (LCS_FMT_C64, device memory); replaces lcs_launch_ingest for these batches
int lcs_launch_ingest_f16(lcs_ctx *c, const void *d_src, int n_buf, uint32_t n_cap) {
  c->src_u8 = false;
  c->src32 = nullptr;
  HIPCHK(c, hipMemsetAsync(c->xmax16, 0, sizeof(unsigned) * n_buf, c->stream));
  if ((n_cap & 1u) == 0 && (reinterpret_cast<uintptr_t>(d_src) & 15u) == 0) {
    // read in place: no copy into cap32; the fp64 stages read the caller's buffers (they stay valid until the batch is
    // collected, include/lcs.h), the maximum is a read-only pass
    c->src32 = static_cast<const float2 *>(d_src);
    const unsigned per_wg = 256 * F16_MAX_ILP, n_part = (n_cap / 2 + per_wg - 1) / per_wg;
    if (n_part > F16_MAXP) { c->err = "capture buffer too long for the fp16 path's partial maxima"; return LCS_ERR_BAD_ARG; }     // > 262144 samples: check_common refuses those
    hipLaunchKernelGGL(k_f16_max4, dim3(n_part, n_buf), dim3(256), 0, c->stream, static_cast<const float4 *>(d_src), n_cap, c->xpart16);
    hipLaunchKernelGGL(k_f16_max_fold, dim3(n_buf), dim3(128), 0, c->stream, c->xpart16, (int)n_part, c->xmax16);
    hipLaunchKernelGGL((k_f16_ingest<true>), dim3((unsigned)((lcs_cap8_stride(n_cap) / 2 + per_wg - 1) / per_wg), n_buf), dim3(256), 0, c->stream, c->src32,
                       n_cap, c->xmax16, c->cap16h, c->cap16l);
  } else {
    int rc = lcs_launch_ingest(c, d_src, LCS_FMT_C64, n_buf, n_cap);
    if (rc) return rc;
    hipLaunchKernelGGL(k_f16_max, dim3(32, n_buf), dim3(256), 0, c->stream, c->cap32, n_cap, c->xmax16);
    hipLaunchKernelGGL((k_f16_ingest<false>), dim3(64, n_buf), dim3(256), 0, c->stream, c->cap32, n_cap, c->xmax16, c->cap16h, c->cap16l);
  }
  HIPCHK(c, hipGetLastError());
  return LCS_OK;
}
int lcs_launch_fill_brow_f16(lcs_ctx *c, int n_buf, const XcGeom &geo) {
  hipLaunchKernelGGL(k_f16_scales, dim3(n_buf), dim3(256), 0, c->stream, c->tmpl, c->xmax16, c->texp16, c->tsc16, geo);
  hipLaunchKernelGGL(k_fill_brow_f16, dim3((LCS_TG * F16R_RLEN + 255) / 256, geo.G, n_buf), dim3(256), 0, c->stream, c->tmpl, c->texp16, c->brow16, geo);
  HIPCHK(c, hipGetLastError());
  return LCS_OK;
}
int lcs_launch_xcorr_f16(lcs_ctx *c, hipStream_t sxc, const XcGeom &geo, int slot0, int n_slots, int xcd_map) {
  const unsigned grid = (unsigned)(F16_TILES * geo.G * n_slots);
  const bool narrow = geo.n_narrow >= geo.n_comb;
  if (narrow)
    hipLaunchKernelGGL(k_xcorr_f16x3<true>, dim3(grid), dim3(256), 0, sxc, c->cap16h, c->cap16l, c->smin, c->start, c->brow16, c->tsc16, c->single, geo,
                       slot0, n_slots, xcd_map);
  else
    hipLaunchKernelGGL(k_xcorr_f16x3<false>, dim3(grid), dim3(256), 0, sxc, c->cap16h, c->cap16l, c->smin, c->start, c->brow16, c->tsc16, c->single, geo,
                       slot0, n_slots, xcd_map);
  HIPCHK(c, hipGetLastError());
  // executed work: per wave and window 3 products x F16_NKB tap blocks x F16_MT sub-tiles x (re, im) MFMAs of 16x16x32 MACs
  const double waves = (double)n_slots * geo.G * ((LCS_N_IDX + F16_MT * 16 - 1) / (F16_MT * 16));      // the waves that own lags (75 of a group's 76)
  c->last_xc_ops += waves * geo.n_comb * (3.0 * (narrow ? F16_NKB - 1 : F16_NKB) * F16_MT * 2) * (2.0 * 16 * 16 * 32);
  c->last_xc_kernel = "k_xcorr_f16x3";
  return LCS_OK;
}
