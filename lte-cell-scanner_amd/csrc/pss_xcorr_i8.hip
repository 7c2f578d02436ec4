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
// Tiling: 256-thread workgroup = 512 output positions x one 16-template group, 8 sub-tiles per wave; per window
// 3 digit passes x 5 tap blocks, fully unrolled.  Digit 2 has its own int32 accumulator; digits 1 and 0 share one
// (shifted left by 8 between the passes), so int -> float conversion happens twice per output and window, not per
// digit.
//
// B operands: resident rows (round 4).  Across the 15 windows of a buffer a template column changes only by its
// delay delta = start[w][foi] - smin[w][g] inside its group.  Rounds 1-3 wrote one 30 KB operand table per (window,
// group) -- 2.8 MB per buffer, written by a fill kernel, streamed through L2 by each of the 19 lag tiles and copied
// into LDS window by window (the kernel's HBM traffic was 1.83 x its algorithmic bytes).  Now a workgroup loads ONE
// image per (buffer, group) at its start (66.5 KB, LDS-DMA) and keeps it for its whole life: per column, digit and
// output the row of (re, im) digit pairs at tap positions -I8R_OFF .. 159 (zeros outside the 137 taps); lane (n, kg)
// reads its 8 taps of tap block kb at dword 16 kb + 4 kg + I8R_OFF / 2 - (delta >> 1) of column n's row.  A pair is 2
// bytes and the delay is any integer: the image holds every row twice, natural (dword i = positions 2i, 2i + 1) and
// shifted by one tap (dword i = positions 2i - 1, 2i), and a lane reads 4 aligned dwords from the copy matching its
// delay's parity -- the scheme of the capture samples.  Rows of consecutive columns sit 88 dwords apart plus n >> 2:
// 88 n = -8 n (mod 32) gives banks {0, 24, 16, 8} + (n >> 2), so the 32 lanes of a dword read (16 columns x 2 tap
// octets, 4 banks apart) fall on 32 different banks when their delays agree.  The per-window LDS-DMA carries only the
// capture samples (6.4 KB of the workgroup's 72.7 KB).  Measured against the table form on one box: kernel alone
// 2.18-2.21 against 2.22-2.27 ms per 128 buffers, in the chain 2.64-2.66 against 2.72-2.74, step + 3.2 %
// (profiles/r04/experiments/ab_i8_resident_rows.txt; a single-copy variant that shifts odd delays into place with
// v_alignbyte_b32 -- 44 KB of LDS -- was 8 % slower, same file).
#include "lcs_internal.h"
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define NW LCS_NW_MAX
#define NFM geo.n_f      // strides of the per-hypothesis / per-group tables: the call's own grid
#define GM geo.G

#ifndef I8_MT
#define I8_MT 8                                          // 16-lag sub-tiles per wave
#endif
#define I8_LAGS (4 * I8_MT * 16)
#define I8_TILES ((LCS_N_IDX + I8_LAGS - 1) / I8_LAGS)
#define I8_NKB LCS_I8_KB                                  // 32-tap blocks per window: 137 taps + delay < 160
// A wave's own staged samples: 128 lags + 32 * I8_NKB taps; lane (i, kg) reads dwords (i + 8 kg) / 2 + 8 S .. + 3 of
// operand S = 2 kb + mt <= 15: 143 dwords per copy.  The shifted copy sits 144 dwords = 16 banks (mod 32) behind the
// natural one, so the even-lag lanes (natural copy, banks 0..12 of a 32-lane group) and the odd-lag lanes (shifted
// copy) of one ds_read never meet on a bank (2 banks apart they did: SQ_LDS_BANK_CONFLICT = 16 % of the kernel's cycles)
#define I8_PCPY 144
static_assert(I8_PCPY >= (15 + 24) / 2 + 8 * (2 * (I8_NKB - 1) + I8_MT - 1) + 4 && I8_PCPY % 32 == 16 && I8_PCPY % 4 == 0, "sample staging");
#define I8_QMAX 8300000.0                                 // |T_int| bound: three balanced base-256 digits reach 8 355 711

// Geometry of the operand image (LCS_I8_OFF, LCS_I8_IMG: lcs_internal.h)
#define I8R_OFF LCS_I8_OFF                                // row position of tap 0; delays stay below it
#define I8R_ROW 88                                        // dwords between the rows of consecutive columns (+ n >> 2)
#define I8R_RLEN 88                                       // dwords of a row: the last one read is 16 * 4 + 4 * 3 + I8R_OFF / 2 + 3
#define I8R_BLK 1412                                      // dwords per (digit, output) block of 16 rows
#define I8R_COPY 8512                                     // dwords per copy: 6 blocks, rounded to whole 64-dword LDS-DMA pieces (and the same banks in both copies)
#define I8R_IMG LCS_I8_IMG
static_assert(I8R_RLEN >= 16 * (I8_NKB - 1) + 12 + I8R_OFF / 2 + 4 && I8R_BLK >= 15 * I8R_ROW + 3 + I8R_RLEN && I8R_COPY >= 6 * I8R_BLK &&
              I8R_COPY % 64 == 0 && I8R_IMG == 2 * I8R_COPY && 137 + I8R_OFF - 1 <= 32 * I8_NKB, "operand image geometry");
__host__ __device__ static inline int i8r_rowoff(int n) { return n * I8R_ROW + (n >> 2); }

// Per template (slot, foi, t): q = I8_QMAX / max tap magnitude; sc = 1 / (128 q) converts the integer
// correlation back to the reference's units.
__global__ __launch_bounds__(256) void k_i8_scales(const float2 *__restrict__ tmpl, double *__restrict__ tq,
                                                   float *__restrict__ sc, XcGeom geo) {
  LCS_TAIL_PRIO();
  const int slot = blockIdx.x;
  for (int e = threadIdx.x; e < geo.G * LCS_TG * 4; e += 256) {      // 4 adjacent lanes per template, 35 taps each
    const int col = e >> 2, qd = e & 3;            // column index: group col / 16, column col % 16
    const int c = lcs_col_tmpl(geo, col >> 4, col & 15);
    float mx = 0.f;
    if (c >= 0) {
      const int foi = c / 3, t = c % 3;
      const float2 *T = tmpl + (((size_t)slot * NFM + foi) * 3 + t) * 137;
      for (int m = qd * 35; m < min(137, qd * 35 + 35); ++m) mx = fmaxf(mx, fmaxf(fabsf(T[m].x), fabsf(T[m].y)));
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1));             // the loop bound is a multiple of 4: a quad is active as a whole
    mx = fmaxf(mx, __shfl_xor(mx, 2));
    if (qd == 0) {
      const double q = (mx > 0.f) ? I8_QMAX / (double)mx : 0.0;
      tq[(size_t)slot * GM * LCS_TG + col] = q;
      sc[(size_t)slot * GM * LCS_TG + col] = (q > 0.0) ? (float)(1.0 / (128.0 * q)) : 0.f;
    }
  }
}

__device__ __forceinline__ void digits3(int v, int &d0, int &d1, int &d2) {   // v = d0 + 256 d1 + 65536 d2, digits in [-128, 127]
  d0 = ((v + 128) & 255) - 128;
  const int v1 = (v - d0) >> 8;          // exact: v - d0 is a multiple of 256
  d1 = ((v1 + 128) & 255) - 128;
  d2 = (v1 - d1) >> 8;
}

// brow[slot][g][copy][digit][op][row(n)][i]: the operand image of one (buffer, group) exactly as it sits in LDS.
// dword i of a natural row = tap positions (2i, 2i + 1), of a shifted row (2i - 1, 2i); position p = tap p - I8R_OFF;
// op 0: pairs (tr, -ti) (real output), op 1: pairs (ti, tr) (imaginary output); digit d of the 24-bit integers.
__global__ __launch_bounds__(256) void k_fill_brow_i8(const float2 *__restrict__ tmpl, const double *__restrict__ tq,
                                                      uint32_t *__restrict__ brow, XcGeom geo) {
  LCS_TAIL_PRIO();
  const int slot = blockIdx.z, g = blockIdx.y;
  uint32_t *out = brow + ((size_t)slot * geo.G + g) * I8R_IMG;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < LCS_TG * 2 * I8R_RLEN; e += gridDim.x * blockDim.x) {
    const int n = e / (2 * I8R_RLEN), r = e % (2 * I8R_RLEN), cpy = r / I8R_RLEN, i = r % I8R_RLEN;
    const int c = lcs_col_tmpl(geo, g, n);
    int tr[2] = {0, 0}, ti[2] = {0, 0};
    if (c >= 0) {
      const int foi = c / 3, t = c % 3;
      const double q = tq[(size_t)slot * GM * LCS_TG + g * LCS_TG + n];      // scales are stored per column
      const float2 *T = tmpl + (((size_t)slot * NFM + foi) * 3 + t) * 137;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int tap = 2 * i - cpy + m - I8R_OFF;
        if (tap >= 0 && tap < 137) { tr[m] = (int)rint((double)T[tap].x * q); ti[m] = (int)rint((double)T[tap].y * q); }
      }
    }
#pragma unroll
    for (int op = 0; op < 2; ++op) {
      uint32_t pk[3] = {0u, 0u, 0u};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = j >> 1;
        const int v = op ? ((j & 1) ? tr[m] : ti[m]) : ((j & 1) ? -ti[m] : tr[m]);
        int d0, d1, d2;
        digits3(v, d0, d1, d2);
        pk[0] |= (uint32_t)(d0 & 255) << (8 * j);
        pk[1] |= (uint32_t)(d1 & 255) << (8 * j);
        pk[2] |= (uint32_t)(d2 & 255) << (8 * j);
      }
#pragma unroll
      for (int d = 0; d < 3; ++d) out[(size_t)cpy * I8R_COPY + (d * 2 + op) * I8R_BLK + i8r_rowoff(n) + i] = pk[d];
    }
  }
}

// The operand image reaches LDS once, at the workgroup's start (one barrier).  From there on a wave works alone: it
// stages its OWN samples -- 128 lags + 160 taps as two copies, natural and shifted by one sample, each a run of dwords
// (= sample pairs) taken from cap8 or cap8s, whichever holds the window start dword aligned: two global_load_lds_dwordx4
// of 36 lanes -- one window ahead, waits only for its own LDS-DMA (s_waitcnt vmcnt(0) at the top of a window), and
// never meets the other three waves at a barrier again (round 4; rounds 1-3 staged one window per workgroup behind a
// barrier per window: same-box A/B + 1 %, 1.8 x the sample bytes through LDS-DMA, 9.2 instead of 6.4 KB of LDS).
// The B operands and the two new A operands of tap block e + 1 are read from LDS behind the first MFMA pair of block e,
// so the LDS latency sits under 14 MFMAs instead of in front of every block.  The digit passes walk the tap blocks
// boustrophedon (digit 2: kb 0..4, digit 1: kb 4..0, digit 0: kb 0..4) so the sliding A window never restarts: 32
// A-operand reads per window instead of 48.
// Epilogue work is spread under the MFMA stream where its inputs allow: the digit-2 sums are converted to float
// while the digit-1 pass runs, the << 8 of the shared digit-1/0 accumulator sits in front of each sub-tile's first
// digit-0 MFMA; what is left behind the last block is 6 VALU operations per output.
// Measured (isolated, 16x16x64 issues every ~18 cycles: tools/microbench/mfma_rate.hip): 240 MFMAs x 18 cycles x 15
// windows x 2 waves per SIMD = 130 k of a workgroup's ~160 k cycles; the rest is the image load, the per-window operand
// latency in front of the first MFMA, and the share of the epilogues the SIMD's other wave does not cover.
__global__ __launch_bounds__(256, 2) void k_xcorr_i8x3(const uint16_t *__restrict__ cap8, const uint16_t *__restrict__ cap8s,
                                                          const int *__restrict__ smin, const int *__restrict__ start,
                                                          const uint32_t *__restrict__ brow,
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

  constexpr int PCPY = I8_PCPY, PBUF = 2 * PCPY;
  __shared__ uint32_t ldsA0[4][PBUF], ldsA1[4][PBUF];      // per wave: [natural copy | shifted copy], two buffers (two VARIABLES, see below)
  __shared__ uint32_t ldsR[I8R_IMG];
  constexpr int NBLK = 3 * I8_NKB;
  const size_t cstride = lcs_cap8_stride(geo.n_cap);
  // (uniform pointers and 32-bit per-lane offsets wherever a lane's address is formed: the kernel sits at its register budget)
  const uint32_t *capd = reinterpret_cast<const uint32_t *>(cap8 + (size_t)slot * cstride);     // dword j = samples (2j, 2j+1)
  const uint32_t *capsd = reinterpret_cast<const uint32_t *>(cap8s + (size_t)slot * cstride);   // dword j = samples (2j+1, 2j+2)
  const int *smin_s = smin + (size_t)slot * NW * GM + g;
  // this lane's column: its delay in window w is start[w][foi] - smin[w][g]; where its row starts in the image
  const int col = lcs_col_tmpl(geo, g, lane & 15);
  const int *start_s = start + (size_t)slot * NW * NFM;
  const int foi_l = (col >= 0) ? col / 3 : 0;
  const int rowbase = i8r_rowoff(lane & 15) + 4 * (lane >> 4) + I8R_OFF / 2;
  {
    // the group's operand image: whole 1 KiB chunks (one global_load_lds_dwordx4 per wave each) + 64-dword pieces
    const uint32_t *img = brow + ((size_t)slot * geo.G + g) * I8R_IMG;
    constexpr int NCI = I8R_IMG / 256, NTL = (I8R_IMG - NCI * 256) / 64;
    static_assert(I8R_IMG == NCI * 256 + NTL * 64 && NTL < 4, "image = whole 1 KiB chunks + up to three 64-dword pieces");
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
  int st_next = start_s[foi_l];              // this lane's window start, fetched one window ahead (I8_DELAY_AND_PREFETCH)
  const int p0 = (lane & 15) + 8 * (lane >> 4);          // relative to the wave's own staged window
  const int par = p0 & 1;
  const int a_dw = (p0 - par) >> 1;

  f32x4 P[I8_MT];
#pragma unroll
  for (int mt = 0; mt < I8_MT; ++mt) P[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // The two sample buffers are two LDS VARIABLES and the window loop is unrolled by two, so that in every window the
  // LDS-DMA destination (the next window's buffer) and the buffer the ds_reads take their operands from are different
  // objects at compile time.  With one array indexed by w & 1 (rounds 1-3) the compiler cannot tell the two halves apart
  // and places s_waitcnt vmcnt(0) in front of the window's first ds_read: every window waited for the NEXT window's
  // samples (and, before the resident rows, its 30 KB of operands) -- the prefetch was never one.  (Removing that wait
  // moved the kernel by < 1 %: the SIMD's other wave covered it.)
  // window W of this wave: samples L0 .. as the natural copy (dword i = samples L0 + 2i, L0 + 2i + 1) and the shifted
  // one (L0 + 2i + 1, L0 + 2i + 2); an odd L0 swaps the roles of cap8 and cap8s
  auto dma = [&](int W, uint32_t (*dst)[PBUF]) __attribute__((always_inline)) {
    const int L0_ = widx0 + smin_s[W * GM], h_ = L0_ >> 1;
    const uint32_t *nat_ = ((L0_ & 1) ? capsd : capd) + h_;
    const uint32_t *shf_ = (L0_ & 1) ? capd + h_ + 1 : capsd + h_;
    if (lane < PCPY / 4) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(nat_ + 4 * lane),
                                       (__attribute__((address_space(3))) void *)(dst[wave]), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(shf_ + 4 * lane),
                                       (__attribute__((address_space(3))) void *)(dst[wave] + PCPY), 16, 0, 0);
    }
  };
#define I8_RD_A(S) { const uint32_t *p_ = bufA + 8 * (S); Aw[S] = (i32x4){(int)p_[0], (int)p_[1], (int)p_[2], (int)p_[3]}; }
  // B operand of (digit D, tap block KB, output OP): 4 dwords of this lane's row in the copy matching its delay's parity
#define I8_RD_B(D, KB, OP, DST)                                                                              \
  {                                                                                                          \
    const uint32_t *q_ = bl + ((D) * 2 + (OP)) * I8R_BLK + 16 * (KB);                                        \
    DST = (i32x4){(int)q_[0], (int)q_[1], (int)q_[2], (int)q_[3]};                                           \
  }
  // behind the wait that opens a window nothing is in flight: the delay fetched during the last window is consumed HERE,
  // before anything new is requested (vmcnt counts in order -- looked at later, the value would drag a wait for the
  // samples requested in between with it); then the next window's start, then its samples (the other buffer: last read
  // in window w - 1)
#define I8_DELAY_AND_PREFETCH                                                                                \
    const int dl = (col >= 0) ? st_next - smin_s[w * GM] : 0;      /* this lane's delay in window w */        \
    const uint32_t *bl = ldsR + (dl & 1) * I8R_COPY + rowbase - (dl >> 1);                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    if (w + 1 < geo.n_comb) { st_next = start_s[(w + 1) * NFM + foi_l]; dma(w + 1, wrA); }                           \
    __builtin_amdgcn_sched_barrier(0);
  auto window = [&](int w, const uint32_t (*rdA)[PBUF], uint32_t (*wrA)[PBUF]) __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);    // vmcnt(0), other counters untouched: this wave's own samples of window w have landed; nobody else reads them
    __builtin_amdgcn_sched_barrier(0);
    I8_DELAY_AND_PREFETCH
    const uint32_t *bufA = rdA[wave] + par * PCPY + a_dw;
    // digit 2 accumulates into (tR, tI); digits 1 and 0 share one int32 accumulator: after the digit-1 pass it is
    // shifted left by 8 and the digit-0 products are added on top (|S1| <= 274 * 128 * 128 = 4.5e6, so
    // 256 S1 + S0 stays below 2^31): one int -> float conversion per digit group instead of per digit.
    i32x4 tR[I8_MT], tI[I8_MT], aR[I8_MT], aI[I8_MT];
    f32x4 fR[I8_MT], fI[I8_MT];            // float(S2): exact, |S2| < 2^24
    i32x4 Aw[2 * I8_NKB + I8_MT - 2];
    i32x4 Bq[2][2];
#pragma unroll
    for (int s = 0; s < I8_MT; ++s) I8_RD_A(s);
#pragma unroll
    for (int op = 0; op < 2; ++op) I8_RD_B(2, 0, op, Bq[0][op]);
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
    for (int e = 0; e < NBLK; ++e) {       // pass d = e / 5 multiplies digit 2 - d
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
#pragma unroll
        for (int op = 0; op < 2; ++op) I8_RD_B(2 - d1, kb1, op, Bq[(e + 1) & 1][op]);
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
  };
  dma(0, ldsA0);
  __syncthreads();                         // the operand image is complete (every wave's chunks: vmcnt(0) in front of the barrier)
  // 9600 lags = 75 wave tiles of 128: the fourth wave of the 19th lag tile owns none.  It has done its share of the image load;
  // nothing below synchronises with it (1 wave in 76: 1.3 % of the launch's MFMAs)
  if (widx0 >= LCS_N_IDX) return;
  for (int w = 0; w < geo.n_comb; w += 2) {
    window(w, ldsA0, ldsA1);
    if (w + 1 < geo.n_comb) window(w + 1, ldsA1, ldsA0);
  }
#undef I8_RD_A
#undef I8_DELAY_AND_PREFETCH
#undef I8_RD_B
  const float ncomb = (float)geo.n_comb;
  const float my_sc = sc[(size_t)slot * GM * LCS_TG + g * LCS_TG + (lane & 15)];
  float *o = sg + (((size_t)slot * geo.G + g) * LCS_N_IDX) * LCS_TG + (lane & 15);
#pragma unroll
  for (int mt = 0; mt < I8_MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int idx = widx0 + mt * 16 + 4 * (lane >> 4) + r;
      if (idx < LCS_N_IDX) o[(size_t)idx * LCS_TG] = __fdiv_rn(P[mt][r] * (my_sc * my_sc), ncomb);
    }
}

int lcs_launch_fill_brow_i8(lcs_ctx *c, int n_buf, const XcGeom &geo) {
  hipLaunchKernelGGL(k_i8_scales, dim3(n_buf), dim3(256), 0, c->stream, c->tmpl, c->tq, c->tsc, geo);
  hipLaunchKernelGGL(k_fill_brow_i8, dim3((LCS_TG * 2 * I8R_RLEN + 255) / 256, geo.G, n_buf), dim3(256), 0, c->stream, c->tmpl, c->tq, c->brow8, geo);
  HIPCHK(c, hipGetLastError());
  return LCS_OK;
}
int lcs_launch_xcorr_i8(lcs_ctx *c, hipStream_t sxc, const XcGeom &geo, int slot0, int n_slots, int xcd_map) {
  const unsigned grid = (unsigned)(I8_TILES * geo.G * n_slots);
  hipLaunchKernelGGL(k_xcorr_i8x3, dim3(grid), dim3(256), 0, sxc, c->cap8, c->cap8s, c->smin, c->start, c->brow8, c->tsc, c->single, geo, slot0,
                     n_slots, xcd_map);
  HIPCHK(c, hipGetLastError());
  // executed work: per wave and window 3 digits x I8_NKB tap blocks x I8_MT sub-tiles x (re, im) MFMAs of 16x16x64 MACs
  const double waves = (double)n_slots * geo.G * ((LCS_N_IDX + I8_MT * 16 - 1) / (I8_MT * 16));      // the waves that own lags (75 of a group's 76)
  c->last_xc_ops += waves * geo.n_comb * (3.0 * I8_NKB * I8_MT * 2) * (2.0 * 16 * 16 * 64);
  c->last_xc_kernel = "k_xcorr_i8x3";
  return LCS_OK;
}
