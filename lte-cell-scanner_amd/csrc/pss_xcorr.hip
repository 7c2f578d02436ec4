// pss_xcorr.hip -- PSS sliding cross-correlation + incoherent combining for gfx950 (MI355X).
//
// Replaces xc_correlate (ref src/searcher.cpp:113-174), xc_combine (:263-308),
// xc_delay_spread (:312-347), sp_est (:185-221) and xc_peak_freq (:353-383).
//
// Design:  the reference materialises xc[3][N-136][n_f] (136 MB at n_f=37)
// and then gathers 15 windows out of it.  Here the 15-window gather is folded into the
// correlation itself: one wavefront owns a tile of 64 output positions (idx) x 16 templates
// (a "group" of consecutive (foi, pss) pairs) and walks the 15 windows; per window it stages
// 64+K' capture samples in LDS (planar, conflict-free), correlates them against the group's
// templates and accumulates |xc|^2 in registers.  The per-foi window start
// round_i(m*.005*k_factor*fs) differs between the templates of a group by a few samples; that
// delay is folded into the template ("B") table, which therefore holds zero-padded, shifted
// copies of conj(fshift(pss_td))/137.  Raw xc never touches HBM.
//
// This file: ingest, tables, signal-power estimate, collapse, and the fp32 correlation kernel
// k_xcorr_mfma_blk, used for sources that are not exact in int8 (complex<float> / complex<double> buffers);
// raw RTL-SDR u8 I/Q takes the int8 three-digit kernel of pss_xcorr_i8.hip.  The complex dot product is
// evaluated as the real product  [xr -xi ; xi xr] x [tr ; ti]  on v_mfma_f32_16x16x4_f32 (exact fp32, a
// k-ordered fma chain): 4-wave workgroups, A operands are Toeplitz slices read straight from LDS planes, B
// operands (template rows) stream through LDS.
#include "lcs_internal.h"
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Internal layout of xc_incoherent_single: sg[slot][g][idx][16] ("group-major"): the 16 templates
// of a group are one 64-byte row per output position, so a correlation wave writes whole
// cache lines and k_collapse reads them back fully coalesced.  The reference layout
// [t][idx][foi] is produced by k_single_to_ref only when a caller asks for that debug output.
#define NW LCS_NW_MAX
// strides of the per-hypothesis / per-group tables: the call's own grid (every call rebuilds its tables: k_prep_tables)
#define NFM geo.n_f
#define GM geo.G

// ------------------------------------------------------------------------------ ingest
// fmt 0: complex<float> in HBM -> cap32; fmt 2: complex<double> already copied into cap64 (host entry points, slot 0
// only) -> cap32 for the fp32 correlation.  For fmt 0 the fp32 copy holds the samples exactly, so no fp64 copy is
// written: the fp64 stages read cap32 and widen on the fly (cap_at()).
__global__ void k_ingest(const void *__restrict__ src, int fmt, uint32_t n_cap, float2 *__restrict__ cap32,
                         const double2 *__restrict__ cap64) {
  LCS_TAIL_PRIO();
  const int slot = blockIdx.y;
  const size_t base = (size_t)slot * n_cap;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_cap; i += gridDim.x * blockDim.x) {
    if (fmt == LCS_FMT_C64) {
      cap32[base + i] = ((const float2 *)src)[base + i];
    } else {
      const double2 v = cap64[base + i];
      cap32[base + i] = make_float2((float)v.x, (float)v.y);
    }
  }
}

// fmt 1: RTL-SDR u8 I/Q, (x-127)/128 (ref src/capbuf.cpp:172-181).  A sample is the integer 127 - u8 (all 256 codes
// fit an int8; the sign is absorbed where the value is used) scaled by -1/128: cap8 gets the (re, im) int8 pair of
// every sample, cap8s the same sequence shifted down by one sample (the correlation kernel's LDS-DMA copies dwords,
// i.e. sample PAIRS: with both phases in memory every window start is dword aligned), both zero-padded behind
// n_cap up to the slot stride.  Nothing wider is stored: the int8 kernel multiplies these bytes, the fp64 stages
// widen them on the fly (cap_at()).  One thread = 8 samples = one 16-byte load and two 16-byte stores.
__global__ __launch_bounds__(256) void k_ingest_u8(const uint8_t *__restrict__ src, uint32_t n_cap, uint16_t *__restrict__ cap8,
                                                   uint16_t *__restrict__ cap8s) {
  LCS_TAIL_PRIO();
  const int slot = blockIdx.y;
  const size_t stride = lcs_cap8_stride(n_cap);
  const uint8_t *in = src + (size_t)slot * n_cap * 2;
  const bool aligned = (((size_t)in) & 15) == 0;
  for (size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i0 < stride; i0 += (size_t)gridDim.x * blockDim.x * 8) {
    uint32_t v[9];      // packed int8 pairs of samples i0 .. i0 + 8
    if (aligned && i0 + 8 <= n_cap) {
      const uint4 q = *reinterpret_cast<const uint4 *>(in + 2 * i0);
      const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t pr = (w[j >> 1] >> (16 * (j & 1))) & 0xffffu;      // (re, im) bytes of sample i0 + j
        v[j] = ((127u - (pr & 255u)) & 255u) | (((127u - (pr >> 8)) & 255u) << 8);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        uint32_t x = 0;
        if (i0 + j < n_cap) { const uchar2 q = reinterpret_cast<const uchar2 *>(in)[i0 + j]; x = ((127u - q.x) & 255u) | (((127u - q.y) & 255u) << 8); }
        v[j] = x;
      }
    }
    {
      uint32_t x = 0;
      if (i0 + 8 < n_cap) { const uchar2 q = reinterpret_cast<const uchar2 *>(in)[i0 + 8]; x = ((127u - q.x) & 255u) | (((127u - q.y) & 255u) << 8); }
      v[8] = x;
    }
    uint4 a, b;
    a.x = v[0] | (v[1] << 16); a.y = v[2] | (v[3] << 16); a.z = v[4] | (v[5] << 16); a.w = v[6] | (v[7] << 16);
    b.x = v[1] | (v[2] << 16); b.y = v[3] | (v[4] << 16); b.z = v[5] | (v[6] << 16); b.w = v[7] | (v[8] << 16);
    *reinterpret_cast<uint4 *>(cap8 + (size_t)slot * stride + i0) = a;
    *reinterpret_cast<uint4 *>(cap8s + (size_t)slot * stride + i0) = b;
  }
}

// fmt 2 with an exactness probe: a complex<double> buffer handed over by a host entry point (already in cap64).  A dongle
// capture holds exactly (u8 - 127) / 128 per component (ref src/capbuf.cpp:172-181; src/LTE-Tracker.cpp:844-845): when
// EVERY component of the buffer is k / 128 with an integer k in [-127, 128], the int8 pairs written here are the ones
// k_ingest_u8 makes from the bytes and the buffer takes the int8 correlation kernel -- which is how the reference's own
// call shape (searcher.h: cvec capbuf) reaches it.  Any other value raises *inexact and the caller correlates the fp32
// copy written in the same pass.  One thread = 8 samples, as k_ingest_u8.
__global__ __launch_bounds__(256) void k_ingest_c128(const double2 *__restrict__ cap64, uint32_t n_cap, float2 *__restrict__ cap32,
                                                     uint16_t *__restrict__ cap8, uint16_t *__restrict__ cap8s, int *__restrict__ inexact) {
  LCS_TAIL_PRIO();
  const size_t stride = lcs_cap8_stride(n_cap);
  bool bad = false;
  for (size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i0 < stride; i0 += (size_t)gridDim.x * blockDim.x * 8) {
    uint32_t v[9];      // packed int8 pairs (127 - u8 = -k) of samples i0 .. i0 + 8
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      uint32_t x = 0;
      if (i0 + j < n_cap) {
        const double2 s = cap64[i0 + j];
        if (j < 8) cap32[i0 + j] = make_float2((float)s.x, (float)s.y);
        const double kr = s.x * 128.0, ki = s.y * 128.0;          // exact: a power of two
        const bool ok = kr == rint(kr) && ki == rint(ki) && kr >= -127.0 && kr <= 128.0 && ki >= -127.0 && ki <= 128.0;
        if (ok) x = ((uint32_t)(-(int)kr) & 255u) | (((uint32_t)(-(int)ki) & 255u) << 8);
        else bad = true;
      }
      v[j] = x;
    }
    uint4 a, b;
    a.x = v[0] | (v[1] << 16); a.y = v[2] | (v[3] << 16); a.z = v[4] | (v[5] << 16); a.w = v[6] | (v[7] << 16);
    b.x = v[1] | (v[2] << 16); b.y = v[3] | (v[4] << 16); b.z = v[5] | (v[6] << 16); b.w = v[7] | (v[8] << 16);
    *reinterpret_cast<uint4 *>(cap8 + i0) = a;
    *reinterpret_cast<uint4 *>(cap8s + i0) = b;
  }
  if (bad) atomicOr(inexact, 1);
}

// ------------------------------------------------------------------------- K0a: tables
// Per slot: window start indices (ref :298), per-(window,group) first offset / tap-pair count,
// and the frequency-shifted conjugated templates (ref :146-151, dsp.h:40-53).
__global__ __launch_bounds__(256) void k_prep_tables(const SlotParams *__restrict__ params,
                                                      const double *__restrict__ fset,
                                                      const double2 *__restrict__ pss_td, float2 *__restrict__ tmpl,
                                                      int *__restrict__ start, int *__restrict__ smin,
                                                      int *__restrict__ kp2, int *__restrict__ n_fix, XcGeom geo) {
  LCS_TAIL_PRIO();
  const int slot = blockIdx.x;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { n_fix[0] = 0; n_fix[1] = 0; }      // this call's list of near-tied positions (k_collapse*, k_frq_repair) and the count of positions left unrepaired
  const SlotParams p = params[slot];
  // the templates (3 x 137 sincos per hypothesis, the long part) are spread over gridDim.y workgroups; the window
  // starts and per-group offsets are few: workgroup y = 0 does them
  for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < geo.n_f * 3 * 137; i += gridDim.y * blockDim.x) {
    const int m = i % 137, t = (i / 137) % 3, foi = i / (137 * 3);
    const double f_off = fset[foi];
    const double kf = (p.fc_req - f_off) / p.fc_prog;
    const double fs = p.fs_prog * kf;
    const double k = M_PI * f_off / (fs / 2);
    const double ang = k * (double)m;
    const double cs = cos(ang), sn = sin(ang);
    const double2 s = pss_td[t * 137 + m];
    const double rr = s.x * cs - s.y * sn, ri = s.x * sn + s.y * cs;   // seq*coeff
    tmpl[(((size_t)slot * NFM + foi) * 3 + t) * 137 + m] = make_float2((float)(rr / 137), (float)(-ri / 137));
  }
  if (blockIdx.y != 0) return;
  // round_i(m*.005*k_factor*fs_programmed), evaluated left to right (ref :298)
  auto win_start = [&](int w, int foi) { return (int)rint((((double)w * .005) * ((p.fc_req - fset[foi]) / p.fc_prog)) * p.fs_prog); };
  for (int i = threadIdx.x; i < geo.n_f * geo.n_comb; i += blockDim.x) {
    const int w = i / geo.n_f, foi = i - w * geo.n_f;
    start[((size_t)slot * NW + w) * NFM + foi] = win_start(w, foi);
  }
  for (int i = threadIdx.x; i < geo.n_comb * geo.G; i += blockDim.x) {
    const int w = i / geo.G, g = i % geo.G;
    const int c_hi = min(g * geo.cpg + geo.cpg - 1, geo.n_tmpl - 1);
    const int f_lo = (g * geo.cpg) / 3, f_hi = c_hi / 3;
    int mn = win_start(w, f_lo), mx = mn;       // (any grid size: recomputed, a group spans at most 7 hypotheses)
    for (int f = f_lo + 1; f <= f_hi; ++f) { const int s = win_start(w, f); mn = min(mn, s); mx = max(mx, s); }
    int k2 = (137 + (mx - mn) + 1) / 2;
    if (k2 > LCS_KP2_MAX - LCS_KP2_UNROLL) k2 = LCS_KP2_MAX - LCS_KP2_UNROLL;   // rejected on the host before launch (lcs_api.hip)
    smin[((size_t)slot * NW + w) * GM + g] = mn;
    kp2[((size_t)slot * NW + w) * GM + g] = k2;
  }
}

// ---------------------------------------------------------------------- K0b: B operands
// btab[slot][w][g][kk][l]:  l = 16*k + j,  k = 0..3 -> (tap 2kk: tr, ti ; tap 2kk+1: tr, ti) of
// template c = 16g + j delayed by start[w][foi(c)] - smin[w][g]; zero outside the 137 taps.
__global__ __launch_bounds__(256) void k_fill_btab(const float2 *__restrict__ tmpl, const int *__restrict__ start,
                                                    const int *__restrict__ smin, const int *__restrict__ kp2,
                                                    float *__restrict__ btab, XcGeom geo, int n_buf) {
  LCS_TAIL_PRIO();
  for (int vb = blockIdx.x; vb < geo.n_comb * geo.G * n_buf; vb += gridDim.x) {     // one (window, group) table of one slot per job
  const int slot = vb / (geo.n_comb * geo.G);
  const int wg = vb % (geo.n_comb * geo.G);
  const int w = wg / geo.G, g = wg % geo.G;
  const int k2 = kp2[((size_t)slot * NW + w) * GM + g];
  const int s0 = smin[((size_t)slot * NW + w) * GM + g];
  float *out = btab + (((size_t)slot * geo.n_comb + w) * geo.G + g) * (size_t)(LCS_KP2_MAX * 64);
  for (int e = threadIdx.x; e < k2 * 64; e += blockDim.x) {
    const int kk = e >> 6, l = e & 63;
    const int c = lcs_col_tmpl(geo, g, l & 15);
    float v = 0.f;
    if (c >= 0) {
      const int foi = c / 3, t = c % 3;
      const int delta = start[((size_t)slot * NW + w) * NFM + foi] - s0;
      const int tap = 2 * kk + (l >> 5) - delta;
      if (tap >= 0 && tap < 137) {
        const float2 T = tmpl[(((size_t)slot * NFM + foi) * 3 + t) * 137 + tap];
        v = ((l >> 4) & 1) ? T.y : T.x;
      }
    }
    out[e] = v;
  }
  }
}

// ---------------------------------------------------------------- K1: correlate+combine
__device__ __forceinline__ float pow2sum(float re, float im) { return fmaf(re, re, im * im); }

// ---------------------------------------------------------------------------------------------
// One 256-thread workgroup owns 4 adjacent lag tiles (256 output positions) of one template group.  The
// capture window is staged once for the four waves, and the template rows (B operands) go through LDS in
// double-buffered chunks of 32 tap pairs instead of being fetched from L2 by every wave (3.6x less L2->CU
// traffic, B-operand latency at LDS level).  One barrier per chunk.  With xcd_map the 1-D grid is laid out so
// that hardware XCD x (observed: workgroup b runs on XCD b % 8) walks slots x, x+8, ... one after the other:
// the 2.9 MB template table and the 1.2 MB capture buffer of a slot then stay in that XCD's 4 MB L2 instead
// of every L2 seeing every slot.  Placement is a speed matter only.
// NWV waves per workgroup (NWV*64 output positions), BCH tap pairs per B chunk.
template <int WPS, int NWV, int BCH>
__global__ __launch_bounds__(NWV * 64, WPS) void k_xcorr_mfma_blk(const float2 *__restrict__ cap32, const int *__restrict__ smin,
                                                              const int *__restrict__ kp2, const float *__restrict__ btab,
                                                              float *__restrict__ sg, XcGeom geo, int slot0, int n_slots,
                                                              int xcd_map) {
  constexpr int NT = NWV * 64;                                   // threads = output positions per workgroup
  constexpr int BLK_LAGS = NT;
  constexpr int BLK_TILES = (LCS_N_IDX + BLK_LAGS - 1) / BLK_LAGS;
  constexpr int PSB = ((BLK_LAGS + 2 * (LCS_KP2_MAX - LCS_KP2_UNROLL) + 31) / 32) * 32 + 16;   // == 16 (mod 32)
  constexpr int ASTEPS = (BLK_LAGS + 2 * (LCS_KP2_MAX - LCS_KP2_UNROLL) + NT - 1) / NT;        // samples per thread per window
  constexpr int BSTEPS = (BCH * 16 + NT - 1) / NT;                                             // float4 per thread per chunk
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per_slot = BLK_TILES * geo.G;
  int q, sidx;
  if (xcd_map) { sidx = blockIdx.x & 7; q = blockIdx.x >> 3; sidx += 8 * (q / per_slot); q = q % per_slot; }
  else { sidx = blockIdx.x / per_slot; q = blockIdx.x % per_slot; }
  if (sidx >= n_slots) return;
  const int slot = slot0 + sidx, g = q / BLK_TILES, idx0 = (q % BLK_TILES) * BLK_LAGS;
  const bool live = idx0 + wave * LCS_LAG_TILE < LCS_N_IDX;      // wave-uniform

  __shared__ float ldsA[2][3 * PSB];
  __shared__ float ldsB[2][BCH * 64];
  const float2 *cap = cap32 + (size_t)slot * geo.n_cap;
  const int *smin_s = smin + (size_t)slot * NW * GM + g;
  const int *kp2_s = kp2 + (size_t)slot * NW * GM + g;
  const int odd = (lane >> 4) & 1;
  const int a1_off = (odd ? 2 * PSB : PSB) + wave * LCS_LAG_TILE + (lane & 15) + (lane >> 5);
  const int a2_off = (odd ? PSB : 0) + wave * LCS_LAG_TILE + (lane & 15) + (lane >> 5);

  f32x4 P[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) P[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // register staging: the capture samples of the next window and the next B chunk
  float2 preA[ASTEPS];
  float4 preB[BSTEPS];
#define BLK_LOAD_A(W)                                                                              \
  {                                                                                                \
    const int L0_ = idx0 + smin_s[(W) * GM], sl_ = BLK_LAGS + 2 * kp2_s[(W) * GM];                 \
    _Pragma("unroll") for (int r_ = 0; r_ < ASTEPS; ++r_) {                                        \
      const int n_ = tid + NT * r_;                                                                \
      const uint32_t s_ = (uint32_t)(L0_ + n_);                                                    \
      preA[r_] = (n_ < sl_ && s_ < geo.n_cap) ? cap[s_] : make_float2(0.f, 0.f);                   \
    }                                                                                              \
  }
#define BLK_LOAD_B(W, C)                                                                           \
  {                                                                                                \
    const float4 *src_ = reinterpret_cast<const float4 *>(                                         \
        btab + (((size_t)slot * geo.n_comb + (W)) * geo.G + g) * (size_t)(LCS_KP2_MAX * 64) + (size_t)(C) * BCH * 64); \
    _Pragma("unroll") for (int r_ = 0; r_ < BSTEPS; ++r_)                                          \
      preB[r_] = ((BCH * 16) % NT == 0 || tid + NT * r_ < BCH * 16) ? src_[tid + NT * r_] : make_float4(0.f, 0.f, 0.f, 0.f); \
  }
  BLK_LOAD_A(0);
  BLK_LOAD_B(0, 0);
  int cb = 0;
  for (int w = 0; w < geo.n_comb; ++w) {
    const int k2 = kp2_s[w * GM];
    const int nch = (k2 + BCH - 1) / BCH;
    float *bufA = ldsA[w & 1];
    {
      const int sl = BLK_LAGS + 2 * k2;
#pragma unroll
      for (int r = 0; r < ASTEPS; ++r) {
        const int n = tid + NT * r;
        if (n < sl) { bufA[n] = preA[r].y; bufA[PSB + n] = preA[r].x; bufA[2 * PSB + n] = -preA[r].y; }
      }
    }
    if (w + 1 < geo.n_comb) BLK_LOAD_A(w + 1);
    f32x4 aR[4], aI[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) { aR[mt] = (f32x4){0.f, 0.f, 0.f, 0.f}; aI[mt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const float *a1p = bufA + a1_off;
    const float *a2p = bufA + a2_off;
    for (int c = 0; c < nch; ++c) {
      float4 *bw = reinterpret_cast<float4 *>(ldsB[cb]);
#pragma unroll
      for (int r = 0; r < BSTEPS; ++r)
        if ((BCH * 16) % NT == 0 || tid + NT * r < BCH * 16) bw[tid + NT * r] = preB[r];
      if (c + 1 < nch) BLK_LOAD_B(w, c + 1)
      else if (w + 1 < geo.n_comb) BLK_LOAD_B(w + 1, 0)
      __syncthreads();
      const float *bl = ldsB[cb] + lane;
      const int nr = min(BCH, k2 - c * BCH);
      const int kbase = c * BCH;
      const int nr4 = nr & ~3;
      int kk = 0;
      for (; kk < nr4; kk += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float b = bl[(kk + u) * 64];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const float a1 = a1p[mt * 16 + 2 * (kbase + kk + u)];
            const float a2 = a2p[mt * 16 + 2 * (kbase + kk + u)];
            aR[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, aR[mt], 0, 0, 0);
            aI[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b, aI[mt], 0, 0, 0);
          }
        }
      }
      for (; kk < nr; ++kk) {
        const float b = bl[kk * 64];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const float a1 = a1p[mt * 16 + 2 * (kbase + kk)];
          const float a2 = a2p[mt * 16 + 2 * (kbase + kk)];
          aR[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, aR[mt], 0, 0, 0);
          aI[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b, aI[mt], 0, 0, 0);
        }
      }
      cb ^= 1;
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) P[mt][r] = P[mt][r] + pow2sum(aR[mt][r], aI[mt][r]);
  }
  if (live) {
    const float ncomb = (float)geo.n_comb;
    float *o = sg + (((size_t)slot * geo.G + g) * LCS_N_IDX) * LCS_TG + (lane & 15);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int idx = idx0 + wave * LCS_LAG_TILE + mt * 16 + 4 * (lane >> 4) + r;
        o[(size_t)idx * LCS_TG] = __fdiv_rn(P[mt][r], ncomb);
      }
  }
}

// ------------------------------------------------------------------------- K2: sp_est
// sp[t] = mean |capbuf[t..t+273]|^2 (ref :204-211), 15-window mean, rotate by 137 (:214-220),
// and the detection threshold of the main loop (ref src/CellSearch.cpp:500-503).  The reference
// updates sp with a serial recurrence; here every window sum is formed directly in fp64 (differs
// from the recurrence at the 1e-14 relative level).
struct SpArgs {
  int n_comb_sp;
  double R_th1, rx_cutoff;
  int n_comb_xc, ds;
};
// grid (tiles of 1024 positions, windows, slots): sp_all[slot][m][i].  Each lane owns 16
// consecutive positions: the 274-sample sum of its first position is formed from 17 sixteen-sample
// segment sums (shared through LDS) plus two samples, then 15 sliding updates (the reference's own
// recurrence, restarted every 16 samples).  LDS index i -> i + i/16 keeps the lanes' 16-sample
// strides on distinct banks.
#define SP_SEG 16
#define SP_TILE (64 * SP_SEG)
__device__ __forceinline__ int sp_pad(int i) { return i + (i >> 4); }
__global__ __launch_bounds__(64) void k_sp_sums(const CapSrc src, double *__restrict__ sp_all, uint32_t n_cap, int n_comb_sp,
                                                 int n_buf) {
  LCS_TAIL_PRIO();
  constexpr int NT = (LCS_N_IDX + SP_TILE - 1) / SP_TILE;
  __shared__ double pw[SP_TILE + 274 + (SP_TILE + 274) / 16 + 2];
  __shared__ double seg[64 + 18];
  const int tid = threadIdx.x;
  for (int vb = blockIdx.x; vb < NT * n_comb_sp * n_buf; vb += gridDim.x) {      // (tile, window, slot), tile fastest
  const int slot = vb / (NT * n_comb_sp), m = (vb / NT) % n_comb_sp;
  const int i0 = (vb % NT) * SP_TILE;
  const CapView cap = cap_view(src, slot);
  __syncthreads();                 // the previous job's readers of pw / seg are done
  const uint32_t base = (uint32_t)m * 9600u + i0;
  constexpr int SP_LD = (SP_TILE + 274 + 63) / 64;     // loads per lane: issued back to back, then consumed
  double2 c[SP_LD];
#pragma unroll
  for (int r = 0; r < SP_LD; ++r) {
    const uint32_t s = base + tid + 64 * r;
    c[r] = (tid + 64 * r < SP_TILE + 274 && s < n_cap) ? cap_at(cap, s) : make_double2(0.0, 0.0);
  }
#pragma unroll
  for (int r = 0; r < SP_LD; ++r) {
    const int n = tid + 64 * r;
    if (n < SP_TILE + 274) pw[sp_pad(n)] = c[r].x * c[r].x + c[r].y * c[r].y;
  }
  __syncthreads();
  const int b0 = tid * SP_SEG;
  for (int k = tid; k < 64 + 17; k += 64) {     // segment sums of the tile and of the 272 samples behind it
    double a = 0;
#pragma unroll
    for (int j = 0; j < SP_SEG; ++j) a += pw[sp_pad(k * SP_SEG + j)];
    seg[k] = a;
  }
  __syncthreads();
  double s = 0;
#pragma unroll
  for (int k = 0; k < 17; ++k) s += seg[tid + k];
  s = s + (pw[sp_pad(b0 + 272)] + pw[sp_pad(b0 + 273)]);
  double *o = sp_all + ((size_t)slot * n_comb_sp + m) * 9600;
  double res[SP_SEG];
#pragma unroll
  for (int q = 0; q < SP_SEG; ++q) {
    if (q) s = s + (-pw[sp_pad(b0 + q - 1)] + pw[sp_pad(b0 + q + 273)]);
    res[q] = s / 274;
  }
  __syncthreads();                 // everyone is done reading pw: reuse it to turn the lane-major results around
#pragma unroll
  for (int q = 0; q < SP_SEG; ++q) pw[sp_pad(b0 + q)] = res[q];
  __syncthreads();
  for (int n = tid; n < SP_TILE; n += 64)
    if (i0 + n < 9600) o[i0 + n] = pw[sp_pad(n)];      // coalesced rows instead of 128-byte-strided stores
  }
}
__global__ __launch_bounds__(256) void k_sp_fold(const double *__restrict__ sp_all, double *__restrict__ spinc,
                                                  double *__restrict__ zth, SpArgs a, int n_buf) {
  LCS_TAIL_PRIO();
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n_buf * 9600; e += gridDim.x * 256) {
    const int slot = e / 9600, i = e % 9600;
    double acc = 0;
    for (int m = 0; m < a.n_comb_sp; ++m) acc += sp_all[((size_t)slot * a.n_comb_sp + m) * 9600 + i];
    const double v = acc / a.n_comb_sp;
    const int o = (i + 137) % 9600;
    spinc[(size_t)slot * 9600 + o] = v;
    zth[(size_t)slot * 9600 + o] = a.R_th1 * v / a.rx_cutoff / 137 / 2 / a.n_comb_xc / (2 * a.ds + 1);
  }
}

// The same for RTL-SDR sources, in integers.  A sample is (127 - u8) / -128, so 16384 |x|^2 is an integer <= 32768 and
// every 274-sample window sum an exact int32: one workgroup owns SPI_TILE positions of one buffer, walks the windows m =
// 0 .. n_comb_sp - 1, forms the prefix sums of the 16384 |x|^2 of SPI_TILE + 274 samples with a block scan and takes
// sp[m][i] = (C[i + 274] - C[i]) / 16384 / 274 -- the same doubles k_sp_sums produces (its fp64 sums of these values
// are exact too) -- accumulated in window order and folded like k_sp_fold.  No per-window array ever reaches HBM.
#define SPI_TILE 960
#define SPI_N (SPI_TILE + 274)
#define SPI_PER ((SPI_N + 255) / 256)
__global__ __launch_bounds__(256) void k_sp_i8(const uint16_t *__restrict__ cap8, uint32_t n_cap, double *__restrict__ spinc,
                                               double *__restrict__ zth, SpArgs a) {
  LCS_TAIL_PRIO();
  __shared__ int C[SPI_N + 1];           // C[k] = sum of the first k powers of the window
  __shared__ int wsum[4];
  const int slot = blockIdx.y, i0 = blockIdx.x * SPI_TILE, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint16_t *cap = cap8 + (size_t)slot * lcs_cap8_stride(n_cap);
  double acc[4] = {0, 0, 0, 0};
  for (int m = 0; m < a.n_comb_sp; ++m) {
    const uint32_t base = (uint32_t)m * 9600u + i0;
    // thread t owns elements t * SPI_PER .. + SPI_PER - 1 (consecutive: a serial prefix, then a scan over the threads)
    int v[SPI_PER], run = 0;
#pragma unroll
    for (int j = 0; j < SPI_PER; ++j) {
      const int k = tid * SPI_PER + j;
      int pw = 0;
      if (k < SPI_N && base + k < n_cap) {
        const uint32_t s = cap[base + k];
        const int re = (int)(int8_t)(s & 255u), im = (int)(int8_t)(s >> 8);
        pw = re * re + im * im;
      }
      run += pw;
      v[j] = run;
    }
    int incl = run;                          // inclusive scan of the per-thread totals: wave, then the 4 waves
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int up = __shfl_up(incl, off); if (lane >= off) incl += up; }
    __syncthreads();                         // previous window's readers of C / wsum are done
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int before = incl - run;
    for (int q = 0; q < wv; ++q) before += wsum[q];
    if (tid == 0) C[0] = 0;
#pragma unroll
    for (int j = 0; j < SPI_PER; ++j) { const int k = tid * SPI_PER + j; if (k < SPI_N) C[k + 1] = before + v[j]; }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = tid + 256 * r;
      if (i < SPI_TILE) acc[r] += ((double)(C[i + 274] - C[i]) / 16384.0) / 274;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + tid + 256 * r;
    if (tid + 256 * r < SPI_TILE && i < 9600) {
      const double vv = acc[r] / a.n_comb_sp;
      const int o = (i + 137) % 9600;
      spinc[(size_t)slot * 9600 + o] = vv;
      zth[(size_t)slot * 9600 + o] = a.R_th1 * vv / a.rx_cutoff / 137 / 2 / a.n_comb_xc / (2 * a.ds + 1);
    }
  }
}

// ------------------------------------------------- K3: delay spread + max over frequency
// ref :312-347 (float adds in the reference's order, circular in idx) and :353-383 (first max).
// Four lanes per output position, one per quad of a group's 16 columns: a lane reads its 16 bytes of the 2*ds+1
// neighbouring 64-byte rows straight from global memory (a wave instruction covers 16 consecutive rows = 1 KB; the
// rows shared with the neighbouring positions come back from L1), sums them in the reference's order and keeps the
// first maximum per PSS over its own columns (ascending in foi); the four lanes of a position then merge their
// candidates -- larger value, on equal values the lower foi: what a single scan in ascending (foi, pss) order with a
// strict comparison keeps.  No LDS, no barrier (round 2 staged the rows in LDS with a 17-float stride whose dword
// writes collided four ways; 85 us alone for 229 MB), 32 VGPRs: the workgroups fit beside two resident correlation
// workgroups (512 - 2 x 232 = 48 VGPRs are free on a SIMD lane).  DS = the arm as a compile-time constant (2: every caller of the reference), or
// -1 = read it from geo.
// Round 5: next to the first maximum the scan keeps the RUNNER-UP value `s` (the largest value of any other hypothesis: the
// middle of {x, best, runner-up} after every column -- one v_med3_f32).  A position whose runner-up lies within
// LCS_FRQ_TIE_EPS of its maximum is a near-tie: the correlation kernels reproduce the reference's values to ~1e-7, not to the
// bit, so there -- and only there -- the arg-max may differ from the reference's.  Such positions go on a list and
// k_frq_repair recomputes their candidates in the reference's own arithmetic.
struct CollapseBest { float v; int foi; float s; };
__device__ __forceinline__ void collapse_step(CollapseBest &b, float xe, int foi) {      // xe = -inf for a column that holds no template
  const bool tk = xe > b.v;
  b.s = __builtin_amdgcn_fmed3f(xe, b.v, b.s);
  b.v = tk ? xe : b.v;
  b.foi = tk ? foi : b.foi;
}
__device__ __forceinline__ void collapse_merge(CollapseBest &b, int lane_xor) {
  const float ov = __shfl_xor(b.v, lane_xor);
  const float os = __shfl_xor(b.s, lane_xor);
  const int of = __shfl_xor(b.foi, lane_xor);
  const bool tk = ov > b.v || (ov == b.v && of < b.foi);
  b.s = fmaxf(fminf(b.v, ov), fmaxf(b.s, os));
  b.v = tk ? ov : b.v;
  b.foi = tk ? of : b.foi;
}
// zth (nullable): the fused single-buffer chains hand out no arrays, only peaks -- and a peak's power is at least its position's
// threshold Z_th1 (ref src/searcher.cpp:449), so there only near-ties that could become a peak are listed (with 0.1 % to spare for
// the power's own ~1e-7): most near-ties sit in the noise, and in latency mode the repair is on the critical path.
__device__ __forceinline__ void collapse_flag(const CollapseBest &r, unsigned pos, unsigned *__restrict__ fix_list, int *__restrict__ n_fix,
                                              const double *__restrict__ zth) {
  if (r.v > 0.f && r.s >= r.v * (1.0f - LCS_FRQ_TIE_EPS)) {
    if (zth && (double)r.v < 0.999 * zth[(pos / (3 * LCS_N_IDX)) * LCS_N_IDX + pos % LCS_N_IDX]) return;
    fix_list[atomicAdd(n_fix, 1)] = pos;
  }
}
typedef const __attribute__((address_space(1))) char *collapse_gptr;
typedef float collapse_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 collapse_row(collapse_gptr base, unsigned byte_off) {   // uniform base + 32-bit lane offset
  const collapse_f4 r = *reinterpret_cast<const __attribute__((address_space(1))) collapse_f4 *>(base + byte_off);
  return make_float4(r.x, r.y, r.z, r.w);
}
template <int DS, bool INCOH>
__global__ __launch_bounds__(256) void k_collapse(const float *__restrict__ sg, float *__restrict__ incoh,
                                                   double *__restrict__ pow_, float *__restrict__ pow32,
                                                   int *__restrict__ frq, unsigned *__restrict__ fix_list, int *__restrict__ n_fix,
                                                   const double *__restrict__ zth, float *__restrict__ second32, XcGeom geo, int n_buf) {
  LCS_TAIL_PRIO();
  constexpr int CT = 64;                           // positions per workgroup
  static_assert(LCS_N_IDX % CT == 0 && LCS_TG == 16, "k_collapse tiles 9600 positions x 16 columns");
  constexpr int NT = LCS_N_IDX / CT;
  constexpr int NA = DS > 0 ? DS : 1;
  const int tid = threadIdx.x, q = tid & 3;
  const int ds = (DS >= 0) ? DS : min(geo.ds, 8);
  const float dsn = (float)(2 * geo.ds + 1);
  for (int vb = blockIdx.x; vb < NT * n_buf; vb += gridDim.x) {       // (position tile, slot), tile fastest
    const int slot = vb / NT;
    const int idx = (vb % NT) * CT + (tid >> 2);
    // byte offsets of this lane's 16 bytes of the rows idx, idx -+ d (circular in idx, ref :336) within one group
    const unsigned o0 = ((unsigned)idx * 4u + q) * 16u;
    unsigned om[NA], op[NA];
#pragma unroll
    for (int d = 1; d <= DS; ++d) {
      om[d - 1] = ((unsigned)((idx - d < 0) ? idx - d + LCS_N_IDX : idx - d) * 4u + q) * 16u;
      op[d - 1] = ((unsigned)((idx + d >= LCS_N_IDX) ? idx + d - LCS_N_IDX : idx + d) * 4u + q) * 16u;
    }
    CollapseBest b0 = {-INFINITY, 0, -INFINITY}, b1 = {-INFINITY, 0, -INFINITY}, b2 = {-INFINITY, 0, -INFINITY};  // any value beats it: foi 0 is always taken
#pragma unroll 1
    for (int g = 0; g < geo.G; ++g) {
      // the group's rows: a wave-uniform base kept in scalar registers (the loads then take base + 32-bit lane offset)
      const uintptr_t gb = reinterpret_cast<uintptr_t>(sg + (((size_t)slot * geo.G + g) * LCS_N_IDX) * LCS_TG);
      const unsigned gb_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(gb >> 32)), gb_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)gb);
      const collapse_gptr rows = reinterpret_cast<collapse_gptr>(((uintptr_t)gb_hi << 32) | (uintptr_t)gb_lo);   // (readfirstlane returns int)
      float4 v = collapse_row(rows, o0);
      if (DS >= 0) {
        float4 a[NA], b[NA];
#pragma unroll
        for (int d = 1; d <= DS; ++d) { a[d - 1] = collapse_row(rows, om[d - 1]); b[d - 1] = collapse_row(rows, op[d - 1]); }
#pragma unroll
        for (int d = 1; d <= DS; ++d) {
          v.x = v.x + (a[d - 1].x + b[d - 1].x); v.y = v.y + (a[d - 1].y + b[d - 1].y);
          v.z = v.z + (a[d - 1].z + b[d - 1].z); v.w = v.w + (a[d - 1].w + b[d - 1].w);
        }
      } else {
        for (int d = 1; d <= ds; ++d) {
          const int im = (idx - d < 0) ? idx - d + LCS_N_IDX : idx - d, ip = (idx + d >= LCS_N_IDX) ? idx + d - LCS_N_IDX : idx + d;
          const float4 a = collapse_row(rows, ((unsigned)im * 4u + q) * 16u), b = collapse_row(rows, ((unsigned)ip * 4u + q) * 16u);
          v.x = v.x + (a.x + b.x); v.y = v.y + (a.y + b.y); v.z = v.z + (a.z + b.z); v.w = v.w + (a.w + b.w);
        }
      }
      // this lane's columns 4q .. 4q+3 of group g are the templates cq .. cq+3 = (foi, pss) in ascending order
      const int cq = g * geo.cpg + 4 * q;
      const int f0 = cq / 3, t0 = cq - 3 * f0;
      float vv[4] = {__fdiv_rn(v.x, dsn), __fdiv_rn(v.y, dsn), __fdiv_rn(v.z, dsn), __fdiv_rn(v.w, dsn)};
      asm volatile("" : "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]));    // all four, here: no per-column branches
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int tj = t0 + j;
        const int t = (tj >= 3) ? tj - 3 : tj, foi = f0 + (tj >= 3 ? 1 : 0);
        const bool valid = (4 * q + j < geo.cpg) && (cq + j < geo.n_tmpl);       // lcs_col_tmpl(geo, g, 4q + j) >= 0
        const float x = vv[j];
        if (INCOH) { if (valid) incoh[((((size_t)slot * 3 + t) * LCS_N_IDX) + idx) * geo.n_f + foi] = x; }
        collapse_step(b0, (valid && t == 0) ? x : -INFINITY, foi);
        collapse_step(b1, (valid && t == 1) ? x : -INFINITY, foi);
        collapse_step(b2, (valid && t == 2) ? x : -INFINITY, foi);
      }
    }
    collapse_merge(b0, 1); collapse_merge(b0, 2);
    collapse_merge(b1, 1); collapse_merge(b1, 2);
    collapse_merge(b2, 1); collapse_merge(b2, 2);
    if (q < 3) {                                                       // lane q of the position writes PSS q
      const CollapseBest r = (q == 0) ? b0 : (q == 1 ? b1 : b2);
      const size_t o = ((size_t)slot * 3 + q) * LCS_N_IDX + idx;
      pow_[o] = (double)r.v;
      pow32[o] = r.v;                                                  // what the fused peak search loads
      frq[o] = r.foi;
      if (second32) second32[o] = r.s;                                 // hypotheses split over GPUs: lcs_foe_contend compares it with the GLOBAL maximum
      collapse_flag(r, (unsigned)o, fix_list, n_fix, zth);
    }
  }
}

// The form every caller of the reference takes (arm 2, 16 columns per group, no debug copy of xc_incoherent): a lane
// loads ONE 16-byte piece per group -- lane = quad * 16 + p, so the 16 lanes of a DPP row hold 16 consecutive positions
// of one quad and the four neighbours of a position arrive by row shifts (v_mov_dpp) instead of four more loads; a wave
// reads positions base-2 .. base+13 and produces the 12 in the middle.  The division by 5 is x * RN(1/5) corrected once
// through the exact remainder (q = x * .2f; q += fma(-5, q, x) * .2f): equal to the correctly rounded quotient for every
// finite float >= 0, denormals included (checked over all 2^31 of them on the host) at 3 operations instead of 12.  With
// 16 columns per group, column j of group g of quad q belongs to PSS (g + q + j) mod 3: with the group loop unrolled by
// three the accumulator of a column is known at compile time in a frame rotated by q, undone once at the end.
template <int CTRL>
__device__ __forceinline__ float collapse_dpp(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float collapse_arm2(float me) {      // me + (idx-1 + idx+1), then + (idx-2 + idx+2)  (ref :336)
  float v = me + (collapse_dpp<0x111>(me) + collapse_dpp<0x101>(me));        // row_shr:1 = from lane p-1, row_shl:1 = from p+1
  return v + (collapse_dpp<0x112>(me) + collapse_dpp<0x102>(me));
}
__device__ __forceinline__ float collapse_div5(float x) {
  const float q = x * 0.2f;
  return fmaf(fmaf(-5.0f, q, x), 0.2f, q);
}
typedef unsigned int collapse_u4 __attribute__((ext_vector_type(4)));
#define COLLAPSE_OUT 12                   // positions a wave produces
#define COLLAPSE_GB 3                     // groups in flight (a multiple of 3; round 4 measured 3 and 6 alike, and the runner-up values need the registers)
// (amdgpu_num_vgpr is doubled by the backend on the unified register file: 28 = the 56 VGPRs that two resident correlation
// workgroups leave free on a SIMD; the allocator otherwise spreads over the 64 its occupancy target allows)
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(28))) void k_collapse_arm2(const float *__restrict__ sg, double *__restrict__ pow_, float *__restrict__ pow32,
                                                        int *__restrict__ frq, unsigned *__restrict__ fix_list, int *__restrict__ n_fix,
                                                        const double *__restrict__ zth, float *__restrict__ second32, XcGeom geo, int n_buf) {
  LCS_TAIL_PRIO();
  constexpr int CT = 4 * COLLAPSE_OUT;             // positions per workgroup
  static_assert(LCS_N_IDX % CT == 0 && LCS_TG == 16, "k_collapse_arm2 tiles 9600 positions x 16 columns");
  constexpr int NT = LCS_N_IDX / CT;
  const int tid = threadIdx.x, lane = tid & 63, p = lane & 15, q = lane >> 4, wv = tid >> 6;
  for (int vb = blockIdx.x; vb < NT * n_buf; vb += gridDim.x) {       // (position tile, slot), tile fastest
    const int slot = vb / NT;
    const int idx = (vb % NT) * CT + wv * COLLAPSE_OUT + p - 2;       // lanes p = 2 .. 13 own an output
    const int ridx = idx < 0 ? idx + LCS_N_IDX : (idx >= LCS_N_IDX ? idx - LCS_N_IDX : idx);      // circular (ref :336)
    const unsigned off = ((unsigned)ridx * 4u + q) * 16u;
    // acc[k]: best of PSS (q + k) mod 3 so far, `foi` holding the COLUMN number 16 g + 4 q + j = 3 foi + pss (ascending
    // with foi within a PSS: the same tie rule; divided by 3 once at the end).  Anything beats -inf.
    CollapseBest acc[3] = {{-INFINITY, 0, -INFINITY}, {-INFINITY, 0, -INFINITY}, {-INFINITY, 0, -INFINITY}};
    // the buffer's groups through one buffer resource: scalar base + per-group scalar offset + the lane's 32-bit offset
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(sg + ((size_t)slot * geo.G * LCS_N_IDX) * LCS_TG), 0, geo.G * (int)(LCS_N_IDX * LCS_TG * sizeof(float)), 0x00020000);
    constexpr int GROUP_BYTES = (int)(LCS_N_IDX * LCS_TG * sizeof(float));
    for (int g6 = 0; g6 < geo.G; g6 += COLLAPSE_GB) {
      // every group of the block is requested before the first one is used: in the gap beside the correlation workgroups
      // only one of these waves fits on a SIMD, and a wave's sequential round trips to memory are its run time
      collapse_u4 me[COLLAPSE_GB];
      if (g6 + COLLAPSE_GB <= geo.G) {                                 // wave-uniform
#pragma unroll
        for (int r = 0; r < COLLAPSE_GB; ++r) me[r] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, (g6 + r) * GROUP_BYTES, 0);
      } else {
#pragma unroll
        for (int r = 0; r < COLLAPSE_GB; ++r) {                        // past the last group the resource returns zeros
          me[r] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, (g6 + r) * GROUP_BYTES, 0);
        }
      }
      const int cb = g6 * 16 + 4 * q;                                  // column number of this lane's first column of the block
#pragma unroll
      for (int r = 0; r < COLLAPSE_GB; ++r) {
        const float mv[4] = {__uint_as_float(me[r].x), __uint_as_float(me[r].y), __uint_as_float(me[r].z), __uint_as_float(me[r].w)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float x = collapse_div5(collapse_arm2(mv[j]));
          const int col = cb + (16 * r + j);
          // the column's PSS is (g + q + j) mod 3 = (q + (r + j)) mod 3: g6 is a multiple of 3
          collapse_step(acc[(r + j) % 3], (col < geo.n_tmpl) ? x : -INFINITY, col);
        }
      }
    }
    // PSS t sits in acc[(t - q) mod 3]; each PSS is merged over the four quads of a position (lanes p, p + 16, p + 32, p + 48)
    const int k0 = (3 - q) % 3, k1 = (4 - q) % 3, k2 = (5 - q) % 3;   // where PSS 0, 1, 2 sit in this lane's frame
    CollapseBest o0 = (k0 == 0) ? acc[0] : (k0 == 1 ? acc[1] : acc[2]);
    CollapseBest o1 = (k1 == 0) ? acc[0] : (k1 == 1 ? acc[1] : acc[2]);
    CollapseBest o2 = (k2 == 0) ? acc[0] : (k2 == 1 ? acc[1] : acc[2]);
    collapse_merge(o0, 16); collapse_merge(o0, 32);
    collapse_merge(o1, 16); collapse_merge(o1, 32);
    collapse_merge(o2, 16); collapse_merge(o2, 32);
    if (q < 3 && p >= 2 && p < 2 + COLLAPSE_OUT) {                     // lane (q, p) writes PSS q of its position
      const CollapseBest rb = (q == 0) ? o0 : (q == 1 ? o1 : o2);
      const float rv = rb.v;
      const int rf = rb.foi / 3;
      int oi = idx;
      asm volatile("" : "+v"(oi));                                     // (the output addresses are formed here, not held across the loop)
      const size_t o = ((size_t)slot * 3 + q) * LCS_N_IDX + oi;
      pow_[o] = (double)rv;
      pow32[o] = rv;                                                   // what the fused peak search loads
      frq[o] = rf;
      if (second32) second32[o] = rb.s;
      collapse_flag(rb, (unsigned)o, fix_list, n_fix, zth);
    }
  }
}

// group-major -> reference layout [t][idx][foi] (debug output of the stage entry point) and back
// (lcs_peak_search receives the reference layout from the caller)
__global__ __launch_bounds__(256) void k_single_to_ref(const float *__restrict__ sg, float *__restrict__ ref, XcGeom geo,
                                                        int to_ref) {
  const int slot = blockIdx.y;
  const size_t n = (size_t)geo.G * LCS_N_IDX * LCS_TG;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(e % LCS_TG);
    const int idx = (int)((e / LCS_TG) % LCS_N_IDX);
    const int g = (int)(e / ((size_t)LCS_TG * LCS_N_IDX));
    const int c = lcs_col_tmpl(geo, g, j);
    if (c < 0) { if (!to_ref) ((float *)sg)[(size_t)slot * n + e] = 0.f; continue; }
    const int foi = c / 3, t = c % 3;
    const size_t r = ((((size_t)slot * 3 + t) * LCS_N_IDX) + idx) * geo.n_f + foi;
    if (to_ref) ref[r] = sg[(size_t)slot * n + e]; else ((float *)sg)[(size_t)slot * n + e] = ref[r];
  }
}

// ---------------------------------------------------------- debug: raw xc (slot 0 only)
// The reference returns xc only "for debugging"; reproduced with its arithmetic (fp64
// accumulate, store as complex<float>) when a caller asks for it.
__global__ __launch_bounds__(256) void k_xc_debug(const double2 *__restrict__ cap64, const SlotParams *__restrict__ params,
                                                   const double *__restrict__ fset, const double2 *__restrict__ pss_td,
                                                   float2 *__restrict__ xc, XcGeom geo) {
  const int foi = blockIdx.y, t = blockIdx.z;
  __shared__ double2 temp[137];
  const SlotParams p = params[0];
  if (threadIdx.x < 137) {
    const int m = threadIdx.x;
    const double f_off = fset[foi];
    const double kf = (p.fc_req - f_off) / p.fc_prog;
    const double fs = p.fs_prog * kf;
    const double k = M_PI * f_off / (fs / 2);
    const double cs = cos(k * (double)m), sn = sin(k * (double)m);
    const double2 s = pss_td[t * 137 + m];
    temp[m] = make_double2((s.x * cs - s.y * sn) / 137, -(s.x * sn + s.y * cs) / 137);
  }
  __syncthreads();
  const uint32_t n_k = geo.n_cap - 136;
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n_k; k += gridDim.x * blockDim.x) {
    double ar = 0, ai = 0;
    for (int m = 0; m < 137; ++m) {
      const double2 a = temp[m], b = cap64[k + m];
      ar += a.x * b.x - a.y * b.y;
      ai += a.x * b.y + a.y * b.x;
    }
    xc[((size_t)t * n_k + k) * geo.n_f + foi] = make_float2((float)ar, (float)ai);
  }
}

// ------------------------------------------------- K3b: near-ties of the arg-max in the reference's own arithmetic
// xc_incoherent_collapsed_frq is an INTEGER output (ref include/searcher.h:31, src/searcher.cpp:353-383): the first maximum over
// the hypotheses of float values that the matrix-core kernels reproduce to ~1e-7 relative, not to the bit.  Wherever the best two
// hypotheses of a (PSS, position) lie within LCS_FRQ_TIE_EPS of each other (k_collapse* list those: a few per buffer) one wave
// recomputes every hypothesis within that distance of the maximum exactly as the reference does --
//   xc[t][k][foi]      fp64 accumulate over the 137 taps in tap order, stored as complex<float>      (:136, :160-169)
//   single += sqr(xc)  the square in double, the running sum a float, window by window, then / n_comb (:299-305)
//   incoherent         float adds  s[i] + (s[i-d] + s[i+d]),  d = 1 .. arm, then / (2 arm + 1)       (:329-345)
// -- and takes the first maximum among them with the reference's strict comparison (:374).  Position, collapsed power
// (then the reference's own float, bit for bit) and index are rewritten in place before the peak search reads them.
__device__ __forceinline__ float repair_gpu_value(const float *__restrict__ sgs, const XcGeom &geo, int c, int idx) {
  const int g = c / geo.cpg, j = c - g * geo.cpg;
  const float *col = sgs + ((size_t)g * LCS_N_IDX) * LCS_TG + j;
  float v = col[(size_t)idx * LCS_TG];
  for (int d = 1; d <= geo.ds; ++d) {
    const int im = (idx - d < 0) ? idx - d + LCS_N_IDX : idx - d, ip = (idx + d >= LCS_N_IDX) ? idx + d - LCS_N_IDX : idx + d;
    v = v + (col[(size_t)im * LCS_TG] + col[(size_t)ip * LCS_TG]);
  }
  return __fdiv_rn(v, (float)(2 * geo.ds + 1));
}
// One 256-thread workgroup per position.  Per candidate the samples all its correlations read -- for every combining window the
// 137 taps + 2 arm lags that follow the window start -- are first staged in LDS by coalesced loads, IN THE SOURCE'S OWN FORMAT
// (2 bytes per sample for dongle data; a lane that walked its own window through global memory met one cache line per tap:
// 140 us per batch for ~300 positions); the 137-term sums then run on LDS operands, one (lag, window) pair per thread.  The
// kernel sits between the collapse and the peak search of every batch: what counts is its latency (a workgroup per position,
// its four waves sharing one candidate's staging and sums; as one wave per position it took 70-120 us).
#define REPAIR_MAX_LAGS 17          // 2 * 8 + 1: lcs_xcorr_pss refuses arms beyond 8
#define REPAIR_SPAN (137 + REPAIR_MAX_LAGS - 1)
#define REPAIR_THREADS 256
#define REPAIR_CROWDED 32          // listed positions per workgroup beyond which the work is bounded (see k_frq_repair)
#define REPAIR_WG_BUDGET 256       // candidates a workgroup recomputes at most once the list is crowded
template <int KIND> struct RepairSample;
template <> struct RepairSample<0> { typedef uint16_t T; static __device__ __forceinline__ double2 cvt(uint16_t p) { return make_double2(-(double)(int)(int8_t)(p & 255u) / 128.0, -(double)(int)(int8_t)(p >> 8) / 128.0); } };
template <> struct RepairSample<1> { typedef float2 T; static __device__ __forceinline__ double2 cvt(float2 f) { return make_double2((double)f.x, (double)f.y); } };
template <> struct RepairSample<2> { typedef double2 T; static __device__ __forceinline__ double2 cvt(double2 d) { return d; } };
// SPLIT (lcs_foe_contend: one buffer's hypotheses split over GPUs): the candidates are this rank's hypotheses within the distance
// of the GLOBAL maximum (from the all-reduced words) plus the global winner itself, whoever owns it -- every rank holds the whole
// buffer and the whole grid (fset = the GLOBAL grid here, this rank's hypotheses sit at foi0 ..) -- and the result goes into a
// second word array that the caller MAX-all-reduces: the exact first maximum over every contender of every rank.
template <int KIND, bool SPLIT>
__global__ __launch_bounds__(REPAIR_THREADS) void k_frq_repair(const float *__restrict__ sg, const unsigned *__restrict__ fix_list,
                                                               const int *__restrict__ n_fix, const CapSrc src,
                                                               const SlotParams *__restrict__ params, const double *__restrict__ fset,
                                                               const double2 *__restrict__ pss_td, const int *__restrict__ start,
                                                               double *__restrict__ pow_, float *__restrict__ pow32, int *__restrict__ frq,
                                                               const long long *__restrict__ words, long long *__restrict__ words2,
                                                               const double *__restrict__ zth, int *__restrict__ n_skipped, XcGeom geo) {
  LCS_TAIL_PRIO();
  typedef typename RepairSample<KIND>::T ST;
  __shared__ double2 s_tmpl[137];
  __shared__ ST s_smp[LCS_NW_MAX][REPAIR_SPAN + 1];
  __shared__ double s_sq[REPAIR_MAX_LAGS * LCS_NW_MAX];
  __shared__ float s_lag[REPAIR_MAX_LAGS];
  const int tid = threadIdx.x, lane = tid & 63;
  const ST *capbase = KIND == 0 ? reinterpret_cast<const ST *>(src.c8) : (KIND == 1 ? reinterpret_cast<const ST *>(src.c32) : reinterpret_cast<const ST *>(src.c64));
  const size_t cap_stride = KIND == 0 ? lcs_cap8_stride(src.n_cap) : (size_t)src.n_cap;
  const int n = *n_fix;
  const int n_lag = 2 * geo.ds + 1, span = 137 + n_lag - 1;
  // Bounded work.  Real data list ~2 positions per buffer.  A degenerate input -- duplicated entries of f_search_set
  // make every position an exact tie -- lists all 28800 per buffer: once the list is longer than REPAIR_CROWDED positions per
  // workgroup, only positions that can become a peak (power at or above their Z_th1, ref :449) are recomputed, and a workgroup
  // stops after REPAIR_WG_BUDGET candidates; the rest keep the collapse kernel's arg-max (for exact duplicates that IS the
  // reference's: identical templates give identical values and the first one wins) and are counted in *n_skipped
  // (lcs_last_frq_repair_stats).
  const bool crowded = n > REPAIR_CROWDED * (int)gridDim.x;
  int spent = 0, skipped = 0;
  for (int e = blockIdx.x; e < n; e += gridDim.x) {
    const unsigned pos = fix_list[e];
    const int idx = (int)(pos % LCS_N_IDX), t = (int)((pos / LCS_N_IDX) % 3), slot = (int)(pos / (3 * LCS_N_IDX));
    if (crowded) {
      const float pv = SPLIT ? __uint_as_float((unsigned)(words[pos] >> 32)) : pow32[pos];
      if (spent >= REPAIR_WG_BUDGET || (double)pv < 0.999 * zth[(size_t)slot * LCS_N_IDX + idx]) { ++skipped; continue; }
    }
    const float *sgs = sg + (size_t)slot * geo.G * LCS_N_IDX * LCS_TG;
    const SlotParams p = params[slot];
    const ST *cap = capbase + (size_t)slot * cap_stride;
    // the values the collapse kernel compared (same expression, same rounding), 64 hypotheses per pass; every wave computes the
    // same candidate masks (no exchange needed)
    const int n_ch = (geo.n_f + 63) >> 6;
    float mx = -INFINITY;
    for (int ch = 0; ch < n_ch; ++ch) {
      const int fl = lane + 64 * ch;
      mx = fmaxf(mx, fl < geo.n_f ? repair_gpu_value(sgs, geo, 3 * fl + t, idx) : -INFINITY);
    }
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    int g_win = -1;                                                      // SPLIT: the global winner's index in the whole grid
    if (SPLIT) {
      const long long gw = words[pos];
      mx = __uint_as_float((unsigned)(gw >> 32));
      g_win = (int)(0xFFFFFFFFu - (unsigned)(gw & 0xFFFFFFFFll));
    }
    const float lim = mx * (1.0f - LCS_FRQ_TIE_EPS);
    const bool win_remote = SPLIT && (geo.foi0 < 0 || g_win < geo.foi0 || g_win >= geo.foi0 + geo.n_f);
    float best = -INFINITY;
    int best_f = 0;
    for (int ch = 0; ch < n_ch + (SPLIT ? 1 : 0); ++ch) {               // SPLIT: one more pass for the remote winner
      unsigned long long m;
      if (ch == n_ch) m = win_remote ? 1ull : 0ull;
      else {
        const int fl = lane + 64 * ch;
        const float x = fl < geo.n_f ? repair_gpu_value(sgs, geo, 3 * fl + t, idx) : -INFINITY;
        m = (SPLIT && geo.foi0 < 0) ? 0ull : __ballot(x >= lim);
      }
      while (m) {                                                        // ascending in foi: the reference's scan order
        const int f_loc = ch == n_ch ? -1 : __builtin_ctzll(m) + 64 * ch;          // -1: the remote winner
        m &= m - 1;
        const int f = SPLIT ? (f_loc < 0 ? g_win : geo.foi0 + f_loc) : f_loc;      // index into fset
        ++spent;
        // conj(fshift(pss_td, f_off, fs_programmed * k_factor)) / 137 in double (ref :146-151, dsp.h:40-53)
        const double f_off = fset[f];
        const double kf = (p.fc_req - f_off) / p.fc_prog;
        const double k = M_PI * f_off / ((p.fs_prog * kf) / 2);
        if (tid < 137) {
          double sn, cs;
          sincos(k * (double)tid, &sn, &cs);
          const double2 s = pss_td[t * 137 + tid];
          s_tmpl[tid] = make_double2((s.x * cs - s.y * sn) / 137, -(s.x * sn + s.y * cs) / 137);
        }
        // window starts: from the table (this rank's own hypotheses) or, for a hypothesis of another rank, by k_prep_tables' expression
        const int *st = start + ((size_t)slot * LCS_NW_MAX) * NFM + (SPLIT ? max(f_loc, 0) : f);
        __shared__ int s_st[LCS_NW_MAX];
        if (tid < LCS_NW_MAX) s_st[tid] = (SPLIT && f_loc < 0) ? (int)rint((((double)tid * .005) * kf) * p.fs_prog) : (tid < geo.n_comb ? st[(size_t)tid * NFM] : 0);
        __syncthreads();
        {
          // thread -> sample o of window w, all of a thread's loads in flight together.  Sample o of window w is what the positions
          // idx - arm .. idx + arm read at tap o - lag; the positions are circular in 9600 (ref :336): the run staged here starts at
          // idx - arm (+ 9600 when that is negative) and serves the lags on its side of the wrap, the others read memory directly
          const int n_it = geo.n_comb * span;
          const size_t at0 = (size_t)(idx - geo.ds + (idx - geo.ds < 0 ? LCS_N_IDX : 0));
          for (int i0 = tid; i0 < n_it; i0 += REPAIR_THREADS * 5) {
            ST v[5];
#pragma unroll
            for (int u = 0; u < 5; ++u) {
              const int i = min(i0 + REPAIR_THREADS * u, n_it - 1), w = i / span, o = i - w * span;
              v[u] = cap[at0 + (size_t)o + (size_t)s_st[w]];
            }
#pragma unroll
            for (int u = 0; u < 5; ++u) {
              const int i = i0 + REPAIR_THREADS * u, w = i / span, o = i - w * span;
              if (i < n_it) s_smp[w][o] = v[u];
            }
          }
        }
        __syncthreads();
        for (int it = tid; it < n_lag * geo.n_comb; it += REPAIR_THREADS) {      // (lag, window): one 137-tap correlation each
          const int l = it / geo.n_comb, w = it - l * geo.n_comb;
          double ar = 0, ai = 0;
          const int ii = idx + l - geo.ds;                               // this lag's position before the circular wrap
          const bool staged = (idx - geo.ds < 0) ? (ii < 0) : (ii < LCS_N_IDX);
          if (staged) {
            const ST *x = &s_smp[w][l];
#pragma unroll 8
            for (int mm = 0; mm < 137; ++mm) {           // (the LDS reads of eight taps in flight; the sums stay in tap order)
              const double2 a = s_tmpl[mm], b = RepairSample<KIND>::cvt(x[mm]);
              ar += a.x * b.x - a.y * b.y;
              ai += a.x * b.y + a.y * b.x;
            }
          } else {                                       // the few lags on the other side of the wrap: straight from memory
            const int iw = ii < 0 ? ii + LCS_N_IDX : (ii >= LCS_N_IDX ? ii - LCS_N_IDX : ii);
            const size_t k0 = (size_t)iw + (size_t)s_st[w];
            for (int mm = 0; mm < 137; ++mm) {
              const double2 a = s_tmpl[mm], b = RepairSample<KIND>::cvt(cap[k0 + mm]);
              ar += a.x * b.x - a.y * b.y;
              ai += a.x * b.y + a.y * b.x;
            }
          }
          const float fr = (float)ar, fi = (float)ai;                    // xc is complex<float>
          s_sq[it] = (double)fr * (double)fr + (double)fi * (double)fi;
        }
        __syncthreads();
        if (tid < n_lag) {                                               // the float running sum over the windows, in window order
          float o = 0.f;
          for (int w = 0; w < geo.n_comb; ++w) o = (float)((double)o + s_sq[tid * geo.n_comb + w]);
          s_lag[tid] = __fdiv_rn(o, (float)geo.n_comb);
        }
        __syncthreads();
        float v = s_lag[geo.ds];
        for (int d = 1; d <= geo.ds; ++d) v = v + (s_lag[geo.ds - d] + s_lag[geo.ds + d]);
        v = __fdiv_rn(v, (float)n_lag);
        if (v > best || (SPLIT && v == best && f < best_f)) { best = v; best_f = f; }      // strict: the lowest index wins a tie (ref :374)
        __syncthreads();                                                 // the LDS arrays are rewritten by the next candidate
      }
    }
    if (tid == 0) {
      if (SPLIT) words2[pos] = ((long long)__float_as_uint(best) << 32) | (long long)(0xFFFFFFFFu - (unsigned)best_f);
      else {
        pow_[pos] = (double)best;
        pow32[pos] = best;
        frq[pos] = best_f;
      }
    }
  }
  if (tid == 0 && skipped) atomicAdd(n_skipped, skipped);
}

// ------------------------------------------------- one-window buffers: xc_incoherent_single in the reference's own arithmetic
// A capture of fewer than 2 x 9600 + 236 samples has ONE combining window (n_comb_xc = 1, ref src/searcher.cpp:276): nothing
// averages over windows, and xc_incoherent_single IS |xc|^2 of one 137-tap sum -- exponentially distributed, so a few hundred of
// its 2.7 M values lie 40 dB and more below the mean.  There the matrix-core kernels' fixed-point templates (24-bit integers, fp16
// hi + lo pairs: an absolute floor of ~1e-7 of the buffer's largest value) show as 1e-4 .. 1e-3 RELATIVE, outside the 1e-5 the
// full-length buffers meet on every element (rounds 1-5 documented that as an exception).  Such a buffer costs 0.5 G fp64
// multiply-adds in all: this kernel forms every element as the reference does (k_frq_repair's arithmetic, element for element: the
// template in double (ref :146-151), the 137 terms added in tap order in double, xc stored as complex<float> (:160-169), the
// square in double, the float running sum over the windows, / n_comb (:299-305)) -- bit for bit the oracle's values.
// One workgroup per (chunk of positions, template group, buffer): thread = (position within a tile of 16, column of the group),
// so a tile's 16 x 16 outputs are sixteen contiguous 64-byte rows; the group's sixteen templates sit in LDS (stride 137 double2:
// the sixteen columns of a wave read sixteen different bank quads).
#define EXS_CHUNKS 32
template <int KIND>
__global__ __launch_bounds__(256) void k_single_exact(const CapSrc src, const SlotParams *__restrict__ params, const double *__restrict__ fset,
                                                       const double2 *__restrict__ pss_td, const int *__restrict__ start,
                                                       float *__restrict__ single, XcGeom geo) {
  typedef typename RepairSample<KIND>::T ST;
  __shared__ double2 s_tmpl[LCS_TG][137];
  __shared__ int s_st[LCS_TG][LCS_NW_MAX];
  const int tid = threadIdx.x, j = tid & (LCS_TG - 1), pp = tid / LCS_TG;
  const int g = blockIdx.y, slot = blockIdx.z;
  const ST *capbase = KIND == 0 ? reinterpret_cast<const ST *>(src.c8) : (KIND == 1 ? reinterpret_cast<const ST *>(src.c32) : reinterpret_cast<const ST *>(src.c64));
  const ST *cap = capbase + (size_t)slot * (KIND == 0 ? lcs_cap8_stride(src.n_cap) : (size_t)src.n_cap);
  const SlotParams p = params[slot];
  for (int e = tid; e < LCS_TG * 137; e += 256) {
    const int jj = e / 137, m = e - jj * 137;
    const int c = lcs_col_tmpl(geo, g, jj);
    double2 v = make_double2(0.0, 0.0);
    if (c >= 0) {
      const int foi = c / 3, t = c - 3 * foi;
      const double f_off = fset[foi];
      const double kf = (p.fc_req - f_off) / p.fc_prog;
      const double k = M_PI * f_off / ((p.fs_prog * kf) / 2);
      double sn, cs;
      sincos(k * (double)m, &sn, &cs);
      const double2 s = pss_td[t * 137 + m];
      v = make_double2((s.x * cs - s.y * sn) / 137, -(s.x * sn + s.y * cs) / 137);
    }
    s_tmpl[jj][m] = v;
  }
  for (int e = tid; e < LCS_TG * LCS_NW_MAX; e += 256) {
    const int jj = e / LCS_NW_MAX, w = e - jj * LCS_NW_MAX;
    const int c = lcs_col_tmpl(geo, g, jj);
    s_st[jj][w] = (c >= 0 && w < geo.n_comb) ? start[((size_t)slot * LCS_NW_MAX + w) * NFM + c / 3] : 0;
  }
  __syncthreads();
  const bool live = lcs_col_tmpl(geo, g, j) >= 0;
  float *out = single + ((size_t)slot * geo.G + g) * LCS_N_IDX * LCS_TG;
  const int per = (LCS_N_IDX + EXS_CHUNKS - 1) / EXS_CHUNKS;
  const int i0 = blockIdx.x * per, i1 = min(LCS_N_IDX, i0 + per);
  for (int idx = i0 + pp; idx < i1; idx += 256 / LCS_TG) {
    float o = 0.f;
    if (live) {
      for (int w = 0; w < geo.n_comb; ++w) {
        const ST *x = cap + (size_t)idx + (size_t)s_st[j][w];
        double ar = 0, ai = 0;
#pragma unroll 8
        for (int m = 0; m < 137; ++m) {
          const double2 a = s_tmpl[j][m], b = RepairSample<KIND>::cvt(x[m]);
          ar += a.x * b.x - a.y * b.y;
          ai += a.x * b.y + a.y * b.x;
        }
        const float fr = (float)ar, fi = (float)ai;                      // xc is complex<float>
        o = (float)((double)o + ((double)fr * (double)fr + (double)fi * (double)fi));
      }
      o = __fdiv_rn(o, (float)geo.n_comb);
    }
    out[(size_t)idx * LCS_TG + j] = o;
  }
}

// lcs_foe_contend, first kernel: which positions does this rank contend for?  The owner of the global winner where its own
// runner-up lies within the distance of the maximum; any other rank where its best does.  words2 starts at -1 (below every word).
__global__ __launch_bounds__(256) void k_foe_flag(const long long *__restrict__ words, const float *__restrict__ pow32, const float *__restrict__ second32,
                                                  long long *__restrict__ words2, unsigned *__restrict__ fix_list, int *__restrict__ n_fix, XcGeom geo) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * LCS_N_IDX) return;
  words2[i] = -1;
  if (geo.foi0 < 0) return;                              // a rank without hypotheses
  const long long gw = words[i];
  const float g = __uint_as_float((unsigned)(gw >> 32));
  const int win = (int)(0xFFFFFFFFu - (unsigned)(gw & 0xFFFFFFFFll));
  const bool mine = win >= geo.foi0 && win < geo.foi0 + geo.n_f;
  const float contender = mine ? second32[i] : pow32[i];
  if (g > 0.f && contender >= g * (1.0f - LCS_FRQ_TIE_EPS)) fix_list[atomicAdd(n_fix, 1)] = (unsigned)i;
}
__global__ __launch_bounds__(256) void k_foe_resolve(long long *__restrict__ words, const long long *__restrict__ words2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 3 * LCS_N_IDX && words2[i] >= 0) words[i] = words2[i];
}

// ------------------------------------------------------------------------------ launch
CapSrc lcs_cap_src(const lcs_ctx *c, uint32_t n_cap) {
  CapSrc s{nullptr, nullptr, nullptr, n_cap};
  if (c->cap64_valid) s.c64 = c->cap64;
  else if (c->src_u8) s.c8 = c->cap8;
  else s.c32 = c->src32 ? c->src32 : c->cap32;
  return s;
}

// complex<double> in cap64 (slot 0): fp32 copy + int8 copies + exactness verdict (one small readback: the caller picks
// the correlation kernel from it).  Needs the int8 buffers (ensure_i8 in lcs_api.hip).
int lcs_launch_ingest_c128(lcs_ctx *c, uint32_t n_cap, bool *exact) {
  c->src_u8 = false;
  c->src32 = nullptr;
  HIPCHK(c, hipMemsetAsync(c->d_flag, 0, sizeof(int), c->stream));
  const unsigned nb = (unsigned)((lcs_cap8_stride(n_cap) / 8 + 255) / 256);
  hipLaunchKernelGGL(k_ingest_c128, dim3(nb), dim3(256), 0, c->stream, c->cap64, n_cap, c->cap32, c->cap8, c->cap8s, c->d_flag);
  HIPCHK(c, hipGetLastError());
  int flag = 1;
  HIPCHK(c, hipMemcpyAsync(&flag, c->d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *exact = flag == 0;
  return LCS_OK;
}

int lcs_launch_ingest(lcs_ctx *c, const void *d_src, int fmt, int n_buf, uint32_t n_cap) {
  c->src_u8 = fmt == LCS_FMT_IQ_U8;
  c->src32 = nullptr;
  if (fmt == LCS_FMT_IQ_U8) {
    const unsigned nb = (unsigned)((lcs_cap8_stride(n_cap) / 8 + 255) / 256);
    hipLaunchKernelGGL(k_ingest_u8, dim3(nb, n_buf), dim3(256), 0, c->stream, (const uint8_t *)d_src, n_cap, c->cap8, c->cap8s);
  } else {
    hipLaunchKernelGGL(k_ingest, dim3(128, n_buf), dim3(256), 0, c->stream, d_src, fmt, n_cap, c->cap32, c->cap64);
  }
  HIPCHK(c, hipGetLastError());
  return LCS_OK;
}

// The correlation kernel saturates the matrix pipes on its own; two of them from different
// contexts (streams) running side by side only evict each other's template tables from L2
// (measured: 2 x 8.7 ms overlapped vs 2 x 7.6 ms back to back).  Launches of this one kernel are
// therefore chained through a per-device event, while everything else a context enqueues (the
// latency-bound per-cell stages in particular) is free to overlap the next context's correlation.
#include <mutex>
static std::mutex g_xc_mutex;
static hipEvent_t g_xc_done[64] = {};

// The fp32 kernel's operand tables: one per (slot, window, group), 0.5 MB each -- only contexts that take this kernel pay for them.
int lcs_ensure_btab(lcs_ctx *c) {
  const size_t need = (size_t)c->cap_slots * LCS_NW_MAX * c->cap_G * LCS_KP2_MAX * 64;
  if (need <= c->btab_elems) return LCS_OK;
  if (c->st_open) { c->err = "the fp32 correlation tables cannot be allocated while a stream is open: lcs_stream_close first"; return LCS_ERR_BAD_ARG; }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->btab) (void)hipFree(c->btab);
  c->btab = nullptr; c->btab_elems = 0;
  HIPCHK(c, hipMalloc((void **)&c->btab, need * sizeof(float)));
  c->btab_elems = need;
  return LCS_OK;
}

int lcs_launch_xcorr(lcs_ctx *c, int n_buf, const XcGeom &geo, bool want_incoh, bool time_it) {
  hipLaunchKernelGGL(k_prep_tables, dim3(n_buf, 4), dim3(256), 0, c->stream, c->params, c->fset, c->d_pss_td, c->tmpl,
                     c->start, c->smin, c->kp2, c->n_fix, geo);
  // one combining window: every element in the reference's own arithmetic (k_single_exact); no operand tables needed
  const bool exact_single = geo.n_comb == 1;
  if (exact_single) ;
  else if (c->use_i8) {
    int rc_ = lcs_launch_fill_brow_i8(c, n_buf, geo);
    if (rc_) return rc_;
  } else if (c->use_f16) {
    int rc_ = lcs_launch_fill_brow_f16(c, n_buf, geo);
    if (rc_) return rc_;
  } else {
    { int rc_ = lcs_ensure_btab(c); if (rc_) return rc_; }
    hipLaunchKernelGGL(k_fill_btab, dim3(geo.n_comb * geo.G * n_buf), dim3(256), 0, c->stream, c->tmpl,
                       c->start, c->smin, c->kp2, c->btab, geo, n_buf);
  }
  // The signal-power estimate and the threshold need the capture buffer only, not the correlation: in the batch / single-buffer chains
  // the correlation sits on stream_xc and they run BESIDE it on the main stream, behind the hand-over event (rounds 1-5 enqueued
  // them in front of it: the correlation of a single buffer waited 26 us for them; lcs_search_capbuf 0.377 -> 0.337 ms same-box).
  // The streaming mode's captured chain keeps everything on one stream: as a parallel branch of the graph (fork to stream_xc, join
  // before the collapse) a replay took 0.315 instead of 0.273 ms -- the runtime replays a forked graph over several streams.
  SpArgs a;
  a.n_comb_sp = (int)((geo.n_cap - 136 - 137) / 9600);
  a.n_comb_xc = geo.n_comb;
  a.ds = geo.ds;
  a.R_th1 = lcs_tables::chi2cdf_inv(1 - pow(10.0, -12), 2.0 * geo.n_comb * (2 * geo.ds + 1));
  a.rx_cutoff = (6 * 12 * 15e3 / 2 + 4 * 15e3) / (30720000.0 / 16 / 2);
  auto launch_sp = [&](hipStream_t st) {
    if (c->src_u8 && !c->cap64_valid) {
      hipLaunchKernelGGL(k_sp_i8, dim3(LCS_N_IDX / SPI_TILE, n_buf), dim3(256), 0, st, c->cap8, geo.n_cap, c->spinc, c->zth, a);
    } else {
      hipLaunchKernelGGL(k_sp_sums, dim3(((LCS_N_IDX + SP_TILE - 1) / SP_TILE) * a.n_comb_sp * n_buf), dim3(64), 0,
                         st, lcs_cap_src(c, geo.n_cap), c->sp, geo.n_cap, a.n_comb_sp, n_buf);       // one-wave workgroups
      hipLaunchKernelGGL(k_sp_fold, dim3((n_buf * LCS_N_IDX + 255) / 256), dim3(256), 0, st, c->sp, c->spinc, c->zth, a, n_buf);
    }
  };

  // slots [0, n8) with the XCD-aware mapping, the remainder with the plain one
  const int n8 = (n_buf >= 8) ? (n_buf & ~7) : 0;
  // main stream -> correlation stream hand-off (tables and capture buffer are ready)
  const bool single_stream = c->single_stream;
  hipStream_t sxc = single_stream ? c->stream : c->stream_xc;
  if (!single_stream) {
    HIPCHK(c, hipEventRecord(c->ev_pre, c->stream));
    HIPCHK(c, hipStreamWaitEvent(sxc, c->ev_pre, 0));
    {
      std::lock_guard<std::mutex> lk(g_xc_mutex);
      hipEvent_t &ev = g_xc_done[c->device & 63];
      if (!ev) HIPCHK(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming | LCS_EVENT_NOFENCE));
      else HIPCHK(c, hipStreamWaitEvent(sxc, ev, 0));
    }
    launch_sp(c->stream);
  } else
    launch_sp(c->stream);
  if (time_it) HIPCHK(c, hipEventRecord(c->ev_xc0, sxc));
  int launches = 0;
  c->last_xc_ops = 0;
  c->last_xc_kernel = "k_xcorr_mfma_blk<4,4,32>";
  if (exact_single) {
    const CapSrc cs = lcs_cap_src(c, geo.n_cap);
    const dim3 grid(EXS_CHUNKS, geo.G, n_buf);
    if (cs.c8) hipLaunchKernelGGL(k_single_exact<0>, grid, dim3(256), 0, sxc, cs, c->params, c->fset, c->d_pss_td, c->start, c->single, geo);
    else if (cs.c32) hipLaunchKernelGGL(k_single_exact<1>, grid, dim3(256), 0, sxc, cs, c->params, c->fset, c->d_pss_td, c->start, c->single, geo);
    else hipLaunchKernelGGL(k_single_exact<2>, grid, dim3(256), 0, sxc, cs, c->params, c->fset, c->d_pss_td, c->start, c->single, geo);
    c->last_xc_kernel = "k_single_exact";
    launches = 1;
  }
  for (int part = 0; part < 2 && !exact_single; ++part) {
    const int s0 = part ? n8 : 0, ns = part ? n_buf - n8 : n8;
    if (ns <= 0) continue;
    if (c->use_i8) {                                                            // u8 sources: int8 three-digit kernel
      int rc_ = lcs_launch_xcorr_i8(c, sxc, geo, s0, ns, part ? 0 : 1);
      if (rc_) return rc_;
    } else if (c->use_f16) {                                                    // complex<float> batches: fp16 three-product kernel
      int rc_ = lcs_launch_xcorr_f16(c, sxc, geo, s0, ns, part ? 0 : 1);
      if (rc_) return rc_;
    } else {                                                                    // fp32: 4-wave workgroups, B through LDS
      constexpr int NWV = 4;
      hipLaunchKernelGGL((k_xcorr_mfma_blk<4, NWV, 32>), dim3((unsigned)(((LCS_N_IDX + NWV * 64 - 1) / (NWV * 64)) * geo.G * ns)),
                         dim3(NWV * 64), 0, sxc, c->cap32, c->smin, c->kp2, c->btab, c->single, geo, s0, ns, part ? 0 : 1);
    }
    ++launches;
  }
  if (time_it) { HIPCHK(c, hipEventRecord(c->ev_xc1, sxc)); c->last_xc_launches = launches; }
  if (!single_stream) {
    {
      std::lock_guard<std::mutex> lk(g_xc_mutex);
      HIPCHK(c, hipEventRecord(g_xc_done[c->device & 63], sxc));
    }
    HIPCHK(c, hipEventRecord(c->ev_post, sxc));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_post, 0));
  }
  {
    const dim3 grid((LCS_N_IDX / 64) * n_buf), block(256);
    float *incoh = want_incoh ? c->incoh : nullptr;
    float *pow32 = reinterpret_cast<float *>(c->work);
    const double *zf = c->repair_peaks_only ? c->zth : nullptr;      // fused single-buffer chains: only near-ties that can become a peak
    float *s2 = c->skip_frq_repair ? c->second32 : nullptr;          // lcs_foe_partial: the runner-up values for lcs_foe_contend
    if (geo.ds == 2 && !incoh && geo.cpg == LCS_TG)
      hipLaunchKernelGGL(k_collapse_arm2, dim3((LCS_N_IDX / (4 * COLLAPSE_OUT)) * n_buf), block, 0, c->stream, c->single, c->pow_, pow32, c->frq, c->fix_list, c->n_fix, zf, s2, geo, n_buf);
    else if (geo.ds == 2 && !incoh) hipLaunchKernelGGL((k_collapse<2, false>), grid, block, 0, c->stream, c->single, incoh, c->pow_, pow32, c->frq, c->fix_list, c->n_fix, zf, s2, geo, n_buf);
    else if (!incoh) hipLaunchKernelGGL((k_collapse<-1, false>), grid, block, 0, c->stream, c->single, incoh, c->pow_, pow32, c->frq, c->fix_list, c->n_fix, zf, s2, geo, n_buf);   // any arm, no debug copy
    else hipLaunchKernelGGL((k_collapse<-1, true>), grid, block, 0, c->stream, c->single, incoh, c->pow_, pow32, c->frq, c->fix_list, c->n_fix, zf, s2, geo, n_buf);
    // near-ties of the arg-max, recomputed in the reference's arithmetic (a few positions per buffer; the kernel loops over the list)
    if (!c->skip_frq_repair && geo.n_f > 1) {             // (one hypothesis -- the streaming mode -- has no arg-max to repair: one graph node less)
      const CapSrc cs = lcs_cap_src(c, geo.n_cap);
      const int ng = std::min(512, 8 * n_buf);
#define REPAIR_LAUNCH(KIND) hipLaunchKernelGGL((k_frq_repair<KIND, false>), dim3(ng), dim3(REPAIR_THREADS), 0, c->stream, c->single, c->fix_list, \
                                               c->n_fix, cs, c->params, c->fset, c->d_pss_td, c->start, c->pow_, pow32, c->frq, nullptr, nullptr, c->zth, c->n_fix + 1, geo)
      if (cs.c8) REPAIR_LAUNCH(0);
      else if (cs.c32) REPAIR_LAUNCH(1);
      else REPAIR_LAUNCH(2);
#undef REPAIR_LAUNCH
    }
  }
  HIPCHK(c, hipGetLastError());
  return LCS_OK;
}

// lcs_foe_contend / lcs_foe_resolve (lcs_api.hip): fset_g = the WHOLE grid on the device
int lcs_launch_foe_contend(lcs_ctx *c, const XcGeom &geo, const double *fset_g, const long long *d_words, long long *d_words2) {
  HIPCHK(c, hipMemsetAsync(c->n_fix, 0, 2 * sizeof(int), c->stream));
  float *pow32 = reinterpret_cast<float *>(c->work);
  hipLaunchKernelGGL(k_foe_flag, dim3((3 * LCS_N_IDX + 255) / 256), dim3(256), 0, c->stream, d_words, pow32, c->second32, d_words2, c->fix_list, c->n_fix, geo);
  const CapSrc cs = lcs_cap_src(c, geo.n_cap);
#define CONTEND_LAUNCH(KIND) hipLaunchKernelGGL((k_frq_repair<KIND, true>), dim3(512), dim3(REPAIR_THREADS), 0, c->stream, c->single, c->fix_list, c->n_fix, cs, \
                                                c->params, fset_g, c->d_pss_td, c->start, c->pow_, pow32, c->frq, d_words, d_words2, c->zth, c->n_fix + 1, geo)
  if (cs.c8) CONTEND_LAUNCH(0);
  else if (cs.c32) CONTEND_LAUNCH(1);
  else CONTEND_LAUNCH(2);
#undef CONTEND_LAUNCH
  HIPCHK(c, hipGetLastError());
  return LCS_OK;
}
int lcs_launch_foe_resolve(lcs_ctx *c, long long *d_words, const long long *d_words2) {
  hipLaunchKernelGGL(k_foe_resolve, dim3((3 * LCS_N_IDX + 255) / 256), dim3(256), 0, c->stream, d_words, d_words2);
  HIPCHK(c, hipGetLastError());
  return LCS_OK;
}

int lcs_launch_single_layout(lcs_ctx *c, const XcGeom &geo, int slot, float *ref_layout, int to_ref) {
  float *sg = c->single + (size_t)slot * geo.G * LCS_N_IDX * LCS_TG;      // the kernel sees one slot
  hipLaunchKernelGGL(k_single_to_ref, dim3(256, 1), dim3(256), 0, c->stream, sg, ref_layout, geo, to_ref);
  HIPCHK(c, hipGetLastError());
  return LCS_OK;
}

int lcs_launch_xc_debug(lcs_ctx *c, const XcGeom &geo) {
  hipLaunchKernelGGL(k_xc_debug, dim3(64, geo.n_f, 3), dim3(256), 0, c->stream, c->cap64, c->params, c->fset,
                     c->d_pss_td, c->xc, geo);
  HIPCHK(c, hipGetLastError());
  return LCS_OK;
}
