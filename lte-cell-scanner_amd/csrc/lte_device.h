// lte_device.h -- device-side LTE helpers shared by the per-cell kernels (tfg_mib.hip) and the tracker kernels
// (tracker.hip): complex fp64 arithmetic, the soft demodulator, and the tail of the PBCH decoder.
#pragma once
#include "lcs_internal.h"

struct cd2 { double re, im; };
__host__ __device__ __forceinline__ cd2 mk(double a, double b) { cd2 r; r.re = a; r.im = b; return r; }
// exp(j x): one sincos call (one argument reduction; cos(x) and sin(x) as two calls cost 1.8 x the instructions)
__device__ __forceinline__ cd2 cis(double x) { double s_, c_; sincos(x, &s_, &c_); return mk(c_, s_); }
// exp(j x) for |x| <= 1 (callers state why): Taylor polynomials in x^2 to x^17 / x^18, Horner with fused multiply-adds; absolute
// error <= 1.5e-16 over the interval (checked against extended precision on 2 M points) at 21 instructions instead of ~100
__host__ __device__ __forceinline__ cd2 cis_small(double x) {
  const double z = x * x;
  double s_ = 2.8114572543455208e-15;            // 1/17!
  s_ = fma(s_, z, -7.6471637318198165e-13);      // -1/15!
  s_ = fma(s_, z, 1.6059043836821613e-10);       // 1/13!
  s_ = fma(s_, z, -2.5052108385441719e-08);      // -1/11!
  s_ = fma(s_, z, 2.7557319223985893e-06);       // 1/9!
  s_ = fma(s_, z, -1.9841269841269841e-04);      // -1/7!
  s_ = fma(s_, z, 8.3333333333333332e-03);       // 1/5!
  s_ = fma(s_, z, -1.6666666666666666e-01);      // -1/3!
  s_ = fma(s_, z, 1.0);
  double c_ = -1.5619206968586226e-16;           // -1/18!
  c_ = fma(c_, z, 4.7794773323873853e-14);       // 1/16!
  c_ = fma(c_, z, -1.1470745597729725e-11);      // -1/14!
  c_ = fma(c_, z, 2.0876756987868099e-09);       // 1/12!
  c_ = fma(c_, z, -2.7557319223985888e-07);      // -1/10!
  c_ = fma(c_, z, 2.4801587301587302e-05);       // 1/8!
  c_ = fma(c_, z, -1.3888888888888889e-03);      // -1/6!
  c_ = fma(c_, z, 4.1666666666666664e-02);       // 1/4!
  c_ = fma(c_, z, -0.5);
  c_ = fma(c_, z, 1.0);
  return mk(c_, x * s_);
}
// WRAP of the reference (include/macros.h): x folded into [sm, lg)
__host__ __device__ __forceinline__ double trk_wrap(double x, double sm, double lg) {
  const double k = x - sm, n = lg - sm;
  return ((n == 0) ? k : (k - n * (double)(int)floor(k / n))) + sm;
}
// The interval of k = x + pi in which the floor of WRAP(x, -pi, pi)'s quotient k / n is CERTAINLY fl: 1e-9 n inside the quotient's
// integer bounds (the division is correctly rounded; 1e-9 is six orders of magnitude more than it or the bounds' own rounding can
// move anything).  nf = n * fl as WRAP forms it.  k_trk_prep's walk subtracts nf where k lies inside and runs WRAP as written otherwise
// (tests/test_viterbi_host.py: no k inside an interval whose fl is not WRAP's).
__host__ __device__ __forceinline__ void trk_wrap_certain_interval(double fl, double &nf, double &lo, double &hi) {
  const double n = M_PI - (-M_PI), g = n * 1e-9;
  const bool sane = fl > -1e6 && fl < 1e6;                 // (double)(int) of WRAP is the identity there
  nf = n * (double)(int)(sane ? fl : 0.0);
  lo = sane ? nf + g : INFINITY;                           // not sane (or not finite): no k passes
  hi = (nf + n) - g;
}
// ---- the producer thread's symbol cutter (tracker.hip: k_trk_cut_hits / k_trk_cut_walk; src/producer_thread.cpp:96-131, 196-246).
// Sample n of a buffer whose first sample has timestamp ts0 carries WRAP(ts0 + n step, 0, 19200); the capture of symbol k (counted
// from slot 0 symbol 0 of the stream's first frame) starts at the first sample behind the previous capture whose tdiff =
// WRAP(timestamp - (frame_timing + target_k), -9600, 9600) passes |tdiff| < 0.5 or 0 < tdiff < 3.  A call cuts symbols k0, k0 + 1, ...
// searching from sample pos0 (0, 0 and ts0 = 0 for a buffer cut from its start; a later buffer of the same stream continues with
// the state the previous call returned).  The same double expressions as lte-cell-scanner_amd/tracker.py cut_symbols and
// host/TrackCells.cpp.
struct TrkCutCell { double step, ft, ts0; int normal; long k0, pos0; };
__host__ __device__ __forceinline__ TrkCutCell trk_cut_cell(int cp_type, double frame_timing, double freq_off, double fc_req, double fc_prog, double fs_prog,
                                                            double ts0, long k0, long pos0) {
  TrkCutCell r;
  const double k_factor = (fc_req - freq_off) / fc_prog;
  r.step = (30720000.0 / 16) / (fs_prog * k_factor);
  r.ft = frame_timing;
  r.ts0 = ts0;
  r.normal = cp_type == LCS_CP_NORMAL;
  r.k0 = k0;
  r.pos0 = pos0;
  return r;
}
// target time of symbol k of the stream: the host's fmod chain runs on integers + 10 / 32 and is exact; *cum = ticks since symbol 0's target
__host__ __device__ __forceinline__ double trk_cut_target(const TrkCutCell &q, long k, double *cum) {
  const double c = q.normal ? 960.0 * (double)(k / 7) + 137.0 * (double)(k % 7) : 160.0 * (double)k;
  *cum = c;
  return fmod((q.normal ? 10.0 : 32.0) + c, 19200.0);
}
__host__ __device__ __forceinline__ bool trk_cut_pass(const TrkCutCell &q, long n, double target, double *tdiff) {
  const double ts = trk_wrap(q.ts0 + (double)n * q.step, 0.0, 19200.0);
  const double d = trk_wrap(ts - (q.ft + target), -9600.0, 9600.0);
  *tdiff = d;
  return fabs(d) < 0.5 || (d > 0 && d < 3);
}
// the call's first symbol (k0): the first sample >= pos0 that passes, searched as the host does (up to 25000 samples, the buffer's
// end permitting).  The window (tdiff in (-0.5, 3): under four samples wide) is either open within the first samples from pos0 or
// opens where the timestamp next passes target - 0.5.
__host__ __device__ inline long trk_cut_first(const TrkCutCell &q, uint32_t n_cap, double *late) {
  double cum, d;
  const double t0 = trk_cut_target(q, q.k0, &cum);
  const long end = (long)n_cap - 128, limit = end < q.pos0 + 25000 ? end : q.pos0 + 25000;
  *late = 0.0;
  if (q.step > 0.9 && q.step < 1.1) {
    for (long n = q.pos0; n <= q.pos0 + 7 && n <= limit; ++n) if (trk_cut_pass(q, n, t0, &d)) { *late = d; return n; }
    const double u = trk_wrap((q.ft + t0 - 0.5) - (q.ts0 + (double)(q.pos0 + 4) * q.step), 0.0, 19200.0);      // ticks from sample pos0 + 4 to the opening
    const long ne = q.pos0 + 4 + (long)ceil(u / q.step);
    for (long n = (ne - 3 > q.pos0 + 8 ? ne - 3 : q.pos0 + 8); n <= ne + 3 && n <= limit; ++n) if (trk_cut_pass(q, n, t0, &d)) { *late = d; return n; }
  }
  return -1;      // (nothing before the limit -- or a step outside the range: trk_cut_symbol reports the premise as broken and the cell is walked)
}
// symbol k0 + j by the closed form, given the first symbol's hit (h0, l0).  *h = first sample of the capture or -1 (not found / does not
// fit in the buffer), *lt = its tdiff.  Returns false when the closed form's premise does not hold -- the sample before the hit passes
// too, or the previous symbol's capture (located the same way) had not ended -- and the cell must be walked sample by sample.
__host__ __device__ inline bool trk_cut_symbol(const TrkCutCell &q, uint32_t n_cap, int j, long h0, double l0, long *h, double *lt) {
  bool premise = q.step > 0.9 && q.step < 1.1;      // the candidate ranges assume about one sample per 1.92 MHz tick
  long hk = -1;
  double lk = 0.0;
  if (j == 0) { hk = h0; lk = l0; }
  else if (h0 >= 0) {
    double cum0, cum, d, dp;
    (void)trk_cut_target(q, q.k0, &cum0);
    const double T0 = (double)h0 * q.step - l0;      // the first symbol's target on the buffer's own clock; symbol k's window opens cum - cum0 - 0.5 later
    const double target = trk_cut_target(q, q.k0 + j, &cum);
    const long ne = (long)ceil((T0 + (cum - cum0) - 0.5) / q.step);
    for (long n = (ne - 2 > 0 ? ne - 2 : 0); n <= ne + 2; ++n) if (trk_cut_pass(q, n, target, &d)) { hk = n; lk = d; break; }
    if (hk < 1 || trk_cut_pass(q, hk - 1, target, &dp)) premise = false;
    else {
      double cumq;
      const double tq = trk_cut_target(q, q.k0 + j - 1, &cumq);
      long hq = (j == 1) ? h0 : -1;
      if (j > 1) {
        const long nq = (long)ceil((T0 + (cumq - cum0) - 0.5) / q.step);
        for (long n = (nq - 2 > 0 ? nq - 2 : 0); n <= nq + 2; ++n) if (trk_cut_pass(q, n, tq, &dp)) { hq = n; break; }
      }
      if (hq < 0 || hq + 128 > hk) premise = false;
    }
  }
  const bool fits = hk >= 0 && hk + 128 <= (long)n_cap;
  *h = fits ? hk : -1;
  *lt = fits ? lk : 0.0;
  return premise;
}
// the walk of the host cutters, sample by sample; returns the number of symbols found (entries behind it: -1 / 0)
__host__ __device__ inline int trk_cut_walk(const TrkCutCell &q, uint32_t n_cap, int n_sym, int *hit, double *late) {
  long pos = q.pos0;
  int j = 0;
  for (; j < n_sym && pos + 128 <= (long)n_cap; ++j) {
    double cum, d = 0.0;
    const double target = trk_cut_target(q, q.k0 + j, &cum);
    const long end = (long)n_cap - 128, limit = end < pos + 25000 ? end : pos + 25000;
    long found = -1;
    for (long n = pos; n <= limit; ++n) if (trk_cut_pass(q, n, target, &d)) { found = n; break; }
    if (found < 0) break;
    hit[j] = (int)found; late[j] = d;
    pos = found + 128;
  }
  const int n_found = j;
  for (; j < n_sym; ++j) { hit[j] = -1; late[j] = 0.0; }
  return n_found;
}

// Real calls, not inlined.  An inlined sincos / atan2 expansion (and cis_small's eighteen coefficients) is ~100 instructions whose
// 64-bit literals the compiler materialises in VGPR pairs and hoists out of the kernel's job loop: every inlined copy costs a
// kernel tens of registers for its whole lifetime (round 5 measured: k_tfg 164 -> 126, k_tfoec_est 161 -> 121, k_chan_est
// 155 -> 127 with the calls below -- the difference between one workgroup and two in the slot a retired correlation workgroup
// leaves, tests/test_tables_abi.py).  Same instructions, same values.
__device__ __attribute__((noinline)) static cd2 cis_call(double x) { return cis(x); }
__device__ __attribute__((noinline)) static cd2 cis_small_call(double x) { return cis_small(x); }
// exp(j x) where x is USUALLY small (the phase ramps of a timing offset of a fraction of a sample): the 21-instruction polynomial
// when every lane's |x| <= 1 (1.5e-16, as cis_small), the library's sincos (~150 instructions) for the wave otherwise
__device__ __attribute__((noinline)) static cd2 cis_auto_call(double x) {
  if (__builtin_amdgcn_ballot_w64(!(fabs(x) <= 1.0)) == 0ull) return cis_small(x);
  return cis(x);
}
__device__ __attribute__((noinline)) static double atan2_call(double y, double x) { return atan2(y, x); }
__host__ __device__ __forceinline__ cd2 cadd(cd2 a, cd2 b) { return mk(a.re + b.re, a.im + b.im); }
__host__ __device__ __forceinline__ cd2 csub(cd2 a, cd2 b) { return mk(a.re - b.re, a.im - b.im); }
__host__ __device__ __forceinline__ cd2 cmul(cd2 a, cd2 b) { return mk(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
__host__ __device__ __forceinline__ cd2 cconj(cd2 a) { return mk(a.re, -a.im); }
__host__ __device__ __forceinline__ cd2 cscale(cd2 a, double s) { return mk(a.re * s, a.im * s); }
__host__ __device__ __forceinline__ cd2 cdivr(cd2 a, double s) { return mk(a.re / s, a.im / s); }
__host__ __device__ __forceinline__ double cabs2(cd2 a) { return a.re * a.re + a.im * a.im; }
__device__ __forceinline__ cd2 ld(const double2 *p) { const double2 v = *p; return mk(v.x, v.y); }
__device__ __forceinline__ void st(double2 *p, cd2 v) { *p = make_double2(v.re, v.im); }
// std::complex division for finite operands (libgcc __divdc3 main path)
__device__ __forceinline__ cd2 cdiv(cd2 x, cd2 y) {
  const double a = x.re, b = x.im, c = y.re, d = y.im;
  if (fabs(c) < fabs(d)) {
    const double ratio = c / d, denom = (c * ratio) + d;
    return mk(((a * ratio) + b) / denom, ((b * ratio) - a) / denom);
  }
  const double ratio = d / c, denom = (d * ratio) + c;
  return mk(((b * ratio) + a) / denom, (b - (a * ratio)) / denom);
}
__device__ __forceinline__ int d_round_i(double x) { return (int)rint(x); }
__device__ __forceinline__ int d_imod(int k, int n) { int r = k % n; return r < 0 ? r + n : r; }
__device__ __forceinline__ int cell_n_symb(const lcs_cell &c) { return c.cp_type == LCS_CP_NORMAL ? 7 : (c.cp_type == LCS_CP_EXTENDED ? 6 : -1); }
__device__ __forceinline__ int cell_id(const lcs_cell &c) { return (c.n_id_1 >= 0 && c.n_id_2 >= 0) ? c.n_id_2 + 3 * c.n_id_1 : -1; }
__device__ __forceinline__ int cn_of(int i) { return (i < 36) ? (i - 36) : (i - 35); }

// ---- 128-point transforms, EIGHT windows per wave, sixteen points per lane in registers (round 5) ----------------------
// Rounds 1-4 kept a window in LDS and ran seven radix-2 stages over it, one butterfly per lane and stage: 28.7 KB of LDS
// traffic per window as 16-byte accesses at power-of-two strides (3.2 conflict cycles per LDS instruction on k_tfg, which
// made the kernel LDS-bound: 2.7 us of a whole CU per 8-window job).  Here lane (w, l) = (lane >> 3, lane & 7) holds the
// points n = l + 8 j, j = 0..15, of window w:
//   Y_l[k2] = sum_j x[l + 8 j] W16^(j k2)            16-point transform in registers (fft16)
//   Z_l[k2] = Y_l[k2] W128^(l k2)                    twiddles from a 2 KB table in LDS (fft128_twiddle_table)
//   X[k2 + 16 k1] = sum_l Z_l[k2] W8^(l k1)          across the 8 lanes of a window: a transpose through LDS in two halves
//                                                    (k2 = 0..7, then 8..15: lane l' receives k2 = l' + 8 c for every l), each
//                                                    followed by an 8-point transform in registers (fft8)
// 4 KB of LDS traffic per window, conflict-free: the 8 lanes of a window write 128 contiguous bytes per store (the store's
// lane groups are 8 contiguous lanes); a column is rotated by k2 >> 1 inside its 8 entries and windows sit 68 entries
// apart, so the 16 lanes of a 16-byte read group (MI355X_MICROARCH.md: {0-3, 12-15, 20-27} ...) fall on 64 different banks
// (checked by enumeration).  8.5 KB of LDS per wave: four waves of a workgroup stay below 40 KB, so that the compiler aims at
// four waves per SIMD (128 registers) -- with a 17 KB buffer it saw two, used all 256, and the kernels no longer fitted
// beside ONE resident correlation workgroup (284 registers are free there): k_sss_win waited 1 ms for a whole CU.
#define FFT128_WSTRIDE 68                               // entries (16 B) per window in the transpose buffer
__host__ __device__ __forceinline__ cd2 cmul_mi(cd2 a) { return mk(a.im, -a.re); }                       // a * (-i)
__host__ __device__ __forceinline__ void fft4(cd2 &a0, cd2 &a1, cd2 &a2, cd2 &a3) {                      // forward, in place, natural order
  const cd2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = cmul_mi(csub(a1, a3));
  a0 = cadd(t0, t2); a2 = csub(t0, t2); a1 = cadd(t1, t3); a3 = csub(t1, t3);
}
// x[n], n = 4 n1 + n2 -> X[k], k = k1 + 4 k2: a 4-point transform over n1 per n2, twiddles W16^(n2 k1), a 4-point transform over n2
__host__ __device__ __forceinline__ void fft16(cd2 (&x)[16]) {
  const double C1 = 0.92387953251128673848, S1 = 0.38268343236508978178, R2 = 0.70710678118654752440;   // cos, sin (pi / 8), sqrt(1/2)
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) fft4(x[n2], x[4 + n2], x[8 + n2], x[12 + n2]);      // x[4 k1 + n2] = A_n2[k1]
  // W16^m = (cos(pi m / 8), -sin(pi m / 8)):  n2 k1 in {1, 2, 3, 2, 4, 6, 3, 6, 9}
  x[4 + 1] = cmul(x[4 + 1], mk(C1, -S1));   x[8 + 1] = cmul(x[8 + 1], mk(R2, -R2));    x[12 + 1] = cmul(x[12 + 1], mk(S1, -C1));
  x[4 + 2] = cmul(x[4 + 2], mk(R2, -R2));   x[8 + 2] = cmul_mi(x[8 + 2]);              x[12 + 2] = cmul(x[12 + 2], mk(-R2, -R2));
  x[4 + 3] = cmul(x[4 + 3], mk(S1, -C1));   x[8 + 3] = cmul(x[8 + 3], mk(-R2, -R2));   x[12 + 3] = cmul(x[12 + 3], mk(-C1, S1));
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) fft4(x[4 * k1], x[4 * k1 + 1], x[4 * k1 + 2], x[4 * k1 + 3]);   // x[4 k1 + k2] = X[k1 + 4 k2]
}
__host__ __device__ __forceinline__ cd2 fft16_out(const cd2 (&x)[16], int k) { return x[4 * (k & 3) + (k >> 2)]; }    // X[k] after fft16
// 8 points in place: x[l] -> X[k1] at x[k1]
__host__ __device__ __forceinline__ void fft8(cd2 (&x)[8]) {
  const double R2 = 0.70710678118654752440;
  // l = 2 l1 + l2, k1 = q1 + 4 q2:  B_l2[q1] = sum_l1 x[2 l1 + l2] W4^(l1 q1);  X[q1 + 4 q2] = B_0[q1] + (-1)^q2 W8^q1 B_1[q1]
  fft4(x[0], x[2], x[4], x[6]);
  fft4(x[1], x[3], x[5], x[7]);
  const cd2 b1 = cmul(x[3], mk(R2, -R2)), b2 = cmul_mi(x[5]), b3 = cmul(x[7], mk(-R2, -R2));
  const cd2 e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6], o0 = x[1];
  x[0] = cadd(e0, o0); x[4] = csub(e0, o0);
  x[1] = cadd(e1, b1); x[5] = csub(e1, b1);
  x[2] = cadd(e2, b2); x[6] = csub(e2, b2);
  x[3] = cadd(e3, b3); x[7] = csub(e3, b3);
}
// W128^(l k2) for l = 0..7, k2 = 0..15 at tw[k2 * 8 + l] (one workgroup-wide table; 128 threads fill it)
__device__ __forceinline__ void fft128_twiddle_table(cd2 *tw, int tid, int n_threads) {
  for (int t = tid; t < 128; t += n_threads) { double s_, c_; sincospi(-(double)((t & 7) * (t >> 3)) / 64.0, &s_, &c_); tw[t] = mk(c_, s_); }
}
// The whole transform for this lane's window: in x[j] = point l + 8 j (j = 0..15); out X[(l' + 8 c) + 16 k1] at x[8 c + k1] with
// l' = this lane's l.  tb: this WAVE's transpose buffer (8 * FFT128_WSTRIDE entries), tw: the twiddle table.
__device__ __forceinline__ void fft128_x8(cd2 (&x)[16], cd2 *tb, const cd2 *tw, int lane) {
  const int w = lane >> 3, l = lane & 7;
  fft16(x);
  cd2 *col = tb + w * FFT128_WSTRIDE;
  // the second half's eight values leave x first (the first half's results then overwrite x[0..7] while they wait)
  cd2 zh[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) zh[kk] = cmul(fft16_out(x, 8 + kk), tw[(8 + kk) * 8 + l]);
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    cd2 z = fft16_out(x, kk);
    if (kk) z = cmul(z, tw[kk * 8 + l]);
    col[kk * 8 + ((l + (kk >> 1)) & 7)] = z;
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    if (c) {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) col[kk * 8 + ((l + (kk >> 1)) & 7)] = zh[kk];
    }
    lcs_wave_sync();
    cd2 y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = col[l * 8 + ((i + (l >> 1)) & 7)];              // Z_i[l + 8 c]
    lcs_wave_sync();                                     // the buffer is rewritten by the second half / the wave's next job
    fft8(y);
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) x[8 * c + k1] = y[k1];
  }
}

// ---- one row of RS_DL (ref src/lte_lib.cpp:305-383): the cell-specific reference symbols of the 6 centre resource
// blocks in OFDM symbol `sym` of slot `slot` (t = 0, 1, 2 -> sym 0, 1, n_symb - 3) as 12 (re, im) pairs, and the per-port
// frequency shifts of that symbol (entries of ports without RS there are left untouched: callers pre-fill -1).
// pn_jump: Gold-sequence jump-ahead table (lcs_tables::pn_jump_table(1600 + 2 * (110 - 6)))
__device__ __forceinline__ void rs_dl_row(int slot, int t, int id, int cp_type, int n_symb, const uint32_t *__restrict__ pn_jump,
                                          double *rs_row /*[12][2]*/, double *shift4) {
  const int sym = (t == 2) ? (n_symb - 3) : t;
  const uint32_t n_cp = (cp_type == LCS_CP_NORMAL);
  const uint32_t c_init = (1u << 10) * (7 * (slot + 1) + sym + 1) * (2 * id + 1) + 2 * id + n_cp;
  // registers after Nc + 2 * (N_RB_MAXDL - 6) clocks (bit index of c(2m) for m = 104), by jump-ahead
  uint32_t x1 = pn_jump[31], x2 = 0;
  for (int b = 0; b < 31; ++b) if ((c_init >> b) & 1u) x2 ^= pn_jump[b];
  uint32_t bits = 0;                                // c(208..231)
  for (int i = 0; i < 24; ++i) {
    bits |= ((x1 ^ x2) & 1u) << i;
    const uint32_t n1 = ((x1 >> 3) ^ x1) & 1u;
    const uint32_t n2 = ((x2 >> 3) ^ (x2 >> 2) ^ (x2 >> 1) ^ x2) & 1u;
    x1 = (x1 >> 1) | (n1 << 30);
    x2 = (x2 >> 1) | (n2 << 30);
  }
  const double isq = 1 / pow(2.0, 0.5);
  for (int k = 0; k < 12; ++k) {
    rs_row[2 * k] = isq * (1 - 2 * (int)((bits >> (2 * k)) & 1u));
    rs_row[2 * k + 1] = isq * (1 - 2 * (int)((bits >> (2 * k + 1)) & 1u));
  }
  for (int port = 0; port < 4; ++port) {
    int v = -1;
    if (port == 0 && sym == 0) v = 0;
    else if (port == 0 && sym == n_symb - 3) v = 3;
    else if (port == 1 && sym == 0) v = 3;
    else if (port == 1 && sym == n_symb - 3) v = 0;
    else if (port == 2 && sym == 1) v = 3 * (slot & 1);
    else if (port == 3 && sym == 1) v = 3 + 3 * (slot & 1);
    const bool want = (t == 0 || t == 2) ? (port <= 1) : (port >= 2);
    if (want && v >= 0) shift4[port] = (double)((v + id) % 6);
  }
}

// ---- soft demodulation of one QPSK symbol: exact log-MAP as itpp::Modulator::demodulate_soft_bits (LOGMAP) with
// rx = sym/sqrt(np), channel = 1/sqrt(np), N0 = 1 (ref src/lte_lib.cpp:612-634)
__device__ __forceinline__ double trunc_log(double x) {     // itpp::trunc_log
  if (x == INFINITY) return log(1.79769313486231570815e+308);
  if (x <= 0) return log(2.22507385850720138309e-308);
  return log(x);
}
// The four likelihoods factor over the two bits, e^{-|rx - g s|^2} = e^{-(rx.re -+ g a)^2} e^{-(rx.im -+ g a)^2}, so as long as no
// exponential underflows the log-MAP value is exactly linear, l0 = 4 rx.re g a and l1 = 4 rx.im g a: what the four exps and
// four logs below compute up to their rounding (~1e-13 absolute on values of order 1-1000, the same size as the difference
// between this device's exp / log and the host libm the oracle uses).  What makes the reference's output differ from the
// linear form is underflow (the far hypotheses' exponentials flush to zero: the LLR saturates below its closed-form value,
// and when all four underflow it is exactly 0, ref trunc_log): every symbol whose farthest hypothesis is within range of
// that (|.|^2 >= 600; exp underflows to subnormals at 708, to zero at 745) takes the reference's arithmetic unchanged.
__device__ __forceinline__ void qpsk_llr(cd2 sym, double np, double &l0, double &l1) {
  const double a = 1 / sqrt(2.0);
  const cd2 gain = cdiv(mk(1.0, 0), mk(sqrt(np), 0));
  const cd2 rx = cmul(sym, gain);
  {
    const double ga = gain.re * a, fr = fabs(rx.re) + ga, fi = fabs(rx.im) + ga;
    if (fr * fr + fi * fi < 600.0) {       // false for NaN / inf as well: those take the reference's path
      l0 = 4.0 * rx.re * ga;
      l1 = 4.0 * rx.im * ga;
      return;
    }
  }
  double metric[4];
  for (int j = 0; j < 4; ++j) {
    const cd2 S = mk((j & 2) ? -a : a, (j & 1) ? -a : a);
    metric[j] = exp(-cabs2(csub(rx, cmul(gain, S))) / 1);
  }
  l0 = trunc_log(metric[0] + metric[1]) - trunc_log(metric[2] + metric[3]);
  l1 = trunc_log(metric[0] + metric[2]) - trunc_log(metric[1] + metric[3]);
}

// ---- tail-biting Viterbi of the PBCH decoder (K = 7, G = (133,171,165)o, 40 steps; ref src/lte_lib.cpp:538-551 -> itpp
// decode_tailbite: one trellis pass per start state, end state forced equal, the best end metric wins).
// PASS 1, one trellis per LANE: lane ss runs the trellis that starts in state ss and keeps all 64 path metrics in registers,
// so a step is 32 butterflies of plain fp64 adds with no cross-lane traffic (the lane-per-state form of all 64 trellises moves
// every metric through ds_bpermute: measured LDS-pipe bound, ~100 us of a whole CU per candidate).  Butterfly j reads old
// states 2j, 2j+1 and writes new states j, j+32 INTO THE SAME TWO REGISTERS, so after k steps state s lives in slot
// rotl6(s, k); the slot pattern repeats every 6 steps, which is the unroll depth.  Branch metrics as IT++ forms them
// (Convolutional_Code::calc_metric: the metrics of all 2^n output words of a step are built first, from the LAST
// generator's observation to the first, and a path metric is old + that one value): d(o) = ((+-r2) + (+-r1)) + (+-r0),
// d(~o) = -d(o) exactly; all three generators tap the input bit and the oldest register bit, so a butterfly needs one
// value and its negative -- 4 fp64 adds and 2 minima per butterfly.
// Round 6: pass 1 keeps NO survivors -- only the end metric of each start state is needed to pick the winner (rounds 1-5
// pushed a decision bit per new state into survivor words, 2 of 5 instructions per state, and parked 20 KB of them per
// candidate in LDS: 64 trellises' worth, of which one is ever traced back).  PASS 2 (vit_retrace_*) runs the ONE winning
// trellis again, one STATE per lane (64 lanes, metrics exchanged by two wave shuffles a step), forms the same sums in the same
// order -- so the same metrics and the same decisions, ties included (the lower-numbered predecessor stays on equal metrics,
// as the reference's strict <) -- keeps each state's 40 decisions in a register and traces back from the forced end state.
#define VIT_ROTL6(s, k) ((((s) << (k)) | ((s) >> (6 - (k)))) & 63)
// output word (o0 + 2 o1 + 4 o2) of the transition (input bit b, old state p)
__host__ __device__ __forceinline__ int vit_word(int b, int p) {
  const int g = (b << 6) | p;
  return __builtin_parity(g & 0133) | (__builtin_parity(g & 0171) << 1) | (__builtin_parity(g & 0165) << 2);
}
// FAST (every observation finite: path metrics are finite or +inf, never NaN): the survivor metric is min(m0, m1) (v_min_f64) --
// equal to the select for every such pair up to the SIGN of a zero, which no later comparison sees.  !FAST keeps the
// compare-and-select form, whose NaN behaviour is the reference's `<`.
// the four branch values of a step, d[i] for the output words i = o0 + 2 o1 (o2 = 0); the other four words are their negatives.
// They are the same for every lane (the observations come from LDS): on the device they are moved into SCALAR registers, where
// they cost no vector register and every v_add_f64 reads one as its scalar operand (held in vector registers the compiler
// fetched the observations of many steps ahead: 139 registers beyond the 256 a wave may have).
struct VitBranch { double d[4]; };
__host__ __device__ __forceinline__ VitBranch vit_branch(double r0, double r1, double r2) {
  VitBranch v;
  v.d[0] = (-r2 + -r1) + -r0; v.d[1] = (-r2 + -r1) + r0; v.d[2] = (-r2 + r1) + -r0; v.d[3] = (-r2 + r1) + r0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int i = 0; i < 4; ++i)
    v.d[i] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v.d[i])), __builtin_amdgcn_readfirstlane(__double2loint(v.d[i])));
#endif
  return v;
}
template <int K, bool FAST>
__host__ __device__ __forceinline__ void vit_step(double (&pm)[64], const VitBranch &br) {
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const int p0 = 2 * j, p1 = 2 * j + 1;
    const int sa = VIT_ROTL6(p0, K), sb = VIT_ROTL6(p1, K);
    const double o0 = pm[sa], o1 = pm[sb];
#pragma unroll
    for (int b = 0; b < 2; ++b) {                      // new state j + 32 b <- (input bit b, predecessor p0 / p1)
      const int w0 = vit_word(b, p0), w1 = vit_word(b, p1);
      const double m0 = o0 + ((w0 < 4) ? br.d[w0 & 3] : -br.d[(7 - w0) & 3]);
      const double m1 = o1 + ((w1 < 4) ? br.d[w1 & 3] : -br.d[(7 - w1) & 3]);
#if defined(__HIP_DEVICE_COMPILE__)
      pm[b ? sb : sa] = FAST ? fmin(m0, m1) : ((m1 < m0) ? m1 : m0);
#else
      pm[b ? sb : sa] = (m1 < m0) ? m1 : m0;           // the host twin: the decision's own comparison (C's fmin leaves the zero's sign open)
#endif
    }
#if defined(__HIP_DEVICE_COMPILE__)
    // four butterflies (24 independent fp64 operations) are all the instruction-level parallelism a SIMD can use; left alone the
    // scheduler interleaves all 32 of a step and several steps, and the temporaries of that cost 140 registers more than a wave has
    if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
#endif
  }
}
// all 40 steps of the trellis that starts in state ss: the metric of the forced end state ss
template <bool FAST>
__host__ __device__ __forceinline__ double vit_end_metric(const double *d0, const double *d1, const double *d2, int ss) {
  double pm[64];
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(ss));      // (the 64 initial metrics are formed HERE: hoisted out of a caller's loop as invariants they are 128 registers spilled across it)
#endif
#pragma unroll
  for (int s = 0; s < 64; ++s) pm[s] = (s == ss) ? 0.0 : INFINITY;
#define VIT_S(K, T) vit_step<K, FAST>(pm, vit_branch(d0[T], d1[T], d2[T]))
#pragma unroll 1
  for (int t6 = 0; t6 < 42; t6 += 6) {
    VIT_S(0, t6); VIT_S(1, t6 + 1); VIT_S(2, t6 + 2); VIT_S(3, t6 + 3);
    if (t6 + 4 < 40) { VIT_S(4, t6 + 4); VIT_S(5, t6 + 5); }
  }
#undef VIT_S
  // 40 = 6 * 6 + 4 steps: state s sits in slot rotl6(s, 4)
  double fin = INFINITY;
#pragma unroll
  for (int s = 0; s < 64; ++s) fin = (s == ss) ? pm[VIT_ROTL6(s, 4)] : fin;
  return fin;
}
// true when every one of the 3 x 40 observations is finite (the FAST form is then exact)
__host__ __device__ __forceinline__ bool vit_finite(double x) { return x - x == 0.0; }

// PASS 2: the ONE winning trellis again, one state per lane, with the lane <-> state map of pass 1's registers: after k steps state s
// lives in lane rotl6(s, k).  The butterfly of old states 2j, 2j + 1 then sits in two lanes that differ in ONE bit (bit k mod 6),
// new state j is formed where 2j was and j + 32 where 2j + 1 was: a step is one exchange with the lane across that bit (a DPP
// move for bits 0-1, a swizzle / permute for the others) instead of two gathers from lanes 2n, 2n + 1 (measured: 8.6 us of
// dependent ds_bpermute round trips per candidate).  Same sums as vit_step -- old + one branch value -- so the same metrics and
// the same decisions; each lane keeps the decisions it took (bit t of a 64-bit word) and the traceback follows the map.
#define VIT_ROTR6(s, k) ((((s) >> (k)) | ((s) << (6 - (k)))) & 63)
// what the lane holding old state s does: which new state it forms (input bit b = s & 1, new state (s >> 1) + 32 b), and which of
// the step's four branch values (and sign) the transitions from 2j and 2j + 1 take.  Packed: i0 | i1 << 2 | neg0 << 4 | neg1 << 5 | b << 6.
__host__ __device__ __forceinline__ int vit_lane_desc(int s) {
  const int b = s & 1, p0 = s & ~1, p1 = s | 1;
  const int w0 = vit_word(b, p0), w1 = vit_word(b, p1);
  const int n0 = w0 >= 4, n1 = w1 >= 4;
  return (n0 ? ((7 - w0) & 3) : (w0 & 3)) | ((n1 ? ((7 - w1) & 3) : (w1 & 3)) << 2) | (n0 << 4) | (n1 << 5) | (b << 6);
}
// the survivor this lane forms from its own metric and its partner's; *take1 = the decision (predecessor 2j + 1 only when STRICTLY better)
__host__ __device__ __forceinline__ double vit_lane_step(int desc, double self, double partner, const double (&d)[4], bool *take1) {
  const int i0 = desc & 3, i1 = (desc >> 2) & 3;
  const double s0 = (i0 == 0) ? d[0] : (i0 == 1) ? d[1] : (i0 == 2) ? d[2] : d[3];
  const double s1 = (i1 == 0) ? d[0] : (i1 == 1) ? d[1] : (i1 == 2) ? d[2] : d[3];
  const bool odd = (desc >> 6) & 1;                    // this lane holds old state 2j + 1
  const double o0 = odd ? partner : self, o1 = odd ? self : partner;
  const double m0 = o0 + (((desc >> 4) & 1) ? -s0 : s0);
  const double m1 = o1 + (((desc >> 5) & 1) ? -s1 : s1);
  *take1 = m1 < m0;
  return *take1 ? m1 : m0;
}
// one traceback step: state s was formed at step t in lane rotl6(s, (t + 1) mod 6); its decision bit names the predecessor
#define VIT_TRACE_LANE(s, t) VIT_ROTL6((s), ((t) + 1) % 6)
#define VIT_TRACE_STEP(bits, s, t, word_of_lane) do { (bits) |= (unsigned long long)(((s) >> 5) & 1) << (t); \
                                                      (s) = (((s) << 1) & 63) | (int)(((word_of_lane) >> (t)) & 1ull); } while (0)
// host twin of the wave's retrace (the 64 lanes walked in a loop): decoded bits, and the end metric of state ss for the tests
__host__ inline unsigned long long vit_retrace_host(const double *d0, const double *d1, const double *d2, int ss, double *end_metric) {
  double pm[64], nx[64];
  unsigned long long dec[64];
  for (int l = 0; l < 64; ++l) { pm[l] = (l == ss) ? 0.0 : INFINITY; dec[l] = 0ull; }
  for (int t = 0; t < 40; ++t) {
    const int k = t % 6;
    const VitBranch br = vit_branch(d0[t], d1[t], d2[t]);
    for (int l = 0; l < 64; ++l) {
      bool take1;
      nx[l] = vit_lane_step(vit_lane_desc(VIT_ROTR6(l, k)), pm[l], pm[l ^ (1 << k)], br.d, &take1);
      dec[l] |= (unsigned long long)(take1 ? 1 : 0) << t;
    }
    for (int l = 0; l < 64; ++l) pm[l] = nx[l];
  }
  if (end_metric) *end_metric = pm[VIT_ROTL6(ss, 40 % 6)];
  unsigned long long bits = 0ull;                      // bit t = decoded bit c_est(t)
  int s = ss;
  for (int t = 39; t >= 0; --t) { const int l = VIT_TRACE_LANE(s, t); VIT_TRACE_STEP(bits, s, t, dec[l]); }
  return bits;
}
#if defined(__HIPCC__)
// the wave's form; all 64 lanes call it with the same (wave-uniform) ss; the decoded bits come back on every lane
__device__ __forceinline__ unsigned long long vit_retrace_wave(const double *d0, const double *d1, const double *d2, int ss, int lane) {
  int desc[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) desc[k] = vit_lane_desc(VIT_ROTR6(lane, k));
  double pm = (lane == ss) ? 0.0 : INFINITY;
  unsigned dlo = 0u, dhi = 0u;                         // the decisions this lane took, bit t (mod 32) of the word for step t
#define VIT_R(K, T) do { const VitBranch br = vit_branch(d0[T], d1[T], d2[T]); bool tk;                                     \
                         pm = vit_lane_step(desc[K], pm, __shfl_xor(pm, 1 << (K)), br.d, &tk);                              \
                         if ((T) < 32) dlo |= (tk ? 1u : 0u) << ((T) & 31); else dhi |= (tk ? 1u : 0u) << ((T) & 31); } while (0)
#pragma unroll 1
  for (int t6 = 0; t6 < 36; t6 += 6) { VIT_R(0, t6); VIT_R(1, t6 + 1); VIT_R(2, t6 + 2); VIT_R(3, t6 + 3); VIT_R(4, t6 + 4); VIT_R(5, t6 + 5); }
  VIT_R(0, 36); VIT_R(1, 37); VIT_R(2, 38); VIT_R(3, 39);
#undef VIT_R
#ifdef PH
  PH(6);
#endif
  // traceback on the scalar unit: the state walked is the same on every lane, its decision word comes by v_readlane
  unsigned long long bits = 0ull;
  int s = __builtin_amdgcn_readfirstlane(ss);
  for (int t = 39; t >= 0; --t) {
    const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)(t < 32 ? dlo : dhi), VIT_TRACE_LANE(s, t));
    bits |= (unsigned long long)((s >> 5) & 1) << t;
    s = ((s << 1) & 63) | (int)((w >> (t & 31)) & 1u);
  }
  return bits;
}
#endif
// CRC-16 (x^16+x^12+x^5+1, zero init) of the 24 payload bits against the received 16, with the antenna-port mask
// (ref src/lte_lib.cpp:637-663, src/searcher.cpp:1628-1636)
__host__ __device__ __forceinline__ int pbch_crc_ok(unsigned long long bits, int n_ports) {
  unsigned crc = 0;
  for (int i = 0; i < 24; ++i) {
    const unsigned msb = ((crc >> 15) & 1u) ^ (unsigned)((bits >> i) & 1ull);
    crc = (crc << 1) & 0xffffu;
    if (msb) crc ^= 0x1021u;
  }
  unsigned rx = 0;                                     // received CRC, bit 15 - t = c_est(24 + t)
  for (int t = 0; t < 16; ++t) rx |= (unsigned)((bits >> (24 + t)) & 1ull) << (15 - t);
  if (n_ports == 2) crc ^= 0xffffu;
  else if (n_ports == 4) crc ^= 0x5555u;               // every second bit, t = 1, 3, ... <-> register bits 14, 12, ...
  return (crc == rx) ? 1 : 0;
}

// ---- PBCH decode by ONE wave from the descrambled LLRs (LDS) to the 40 decoded bits: de-ratematch, the 64 tail-biting
// trellises one per lane (end metrics), the winner once more one state per lane (decisions, traceback), CRC-16 with the
// antenna-port mask (ref src/lte_lib.cpp:469-518, 538-551, 637-663; src/searcher.cpp:1617-1636).  e_est: the wave's m_bit LLRs in
// LDS, d_est: 3 x 40 doubles of LDS of the wave's own.  Called by all 64 lanes of ONE wave after the LLR writes; only wave-level
// synchronisation inside (the waves of a workgroup run independent candidates); ok / bits40 are valid on every lane afterwards.
// The trellis pass stays out of line: it takes ~170 registers of its own; inlined, whatever lives across it would spill.
template <bool FAST>
static __device__ __forceinline__ double pbch_trellis_pass(const double (*d_est)[40], int ss) {
  return vit_end_metric<FAST>(d_est[0], d_est[1], d_est[2], ss);
}
__device__ __forceinline__ void pbch_decode_wave(const double *e_est, double (*d_est)[40], const int16_t *__restrict__ derm_inv,
                                                 int m_bit, int n_ports, int lane, int &ok, unsigned long long &bits40) {
  // de-ratematch: average all observations of each coded bit, added in ascending position (ref src/lte_lib.cpp:497-509); the
  // (up to 16) LLRs of a bit are fetched first, all in flight together, and summed in order
  lcs_wave_sync();
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int bit = lane + 64 * q;
    if (bit < 120) {
      const int16_t *lst = derm_inv + ((m_bit == 1920) ? 0 : 120 * 16) + bit * 16;   // ascending bit positions, -1 padded
      double v[16];
      int cnt = 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) { const int t = lst[k]; v[k] = (t >= 0) ? e_est[t] : 0.0; cnt += (t >= 0); }
      double s = 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) if (k < cnt) s += v[k];
      if (cnt > 1) s = s / cnt;
      d_est[bit / 40][bit % 40] = s;
    }
  }
  lcs_wave_sync();
#ifdef PH
  PH(2);
#endif
  // lane = start state; the best end metric wins, the lowest start state among equals (ascending, strict < in the reference).
  // Finite observations (every real capture) take the min form of the step; anything else the compare-and-select form.
  bool fin_in = true;
  for (int k = lane; k < 120; k += 64) fin_in = fin_in && vit_finite(d_est[k / 40][k % 40]);
  const bool finite = !__any(!fin_in);
  const double fin = finite ? pbch_trellis_pass<true>(d_est, lane) : pbch_trellis_pass<false>(d_est, lane);
#ifdef PH
  PH(3);
#endif
  double best = (fin < INFINITY) ? fin : INFINITY;       // NaN / unreachable never win
  int best_ss = lane;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double ov = __shfl_xor(best, off);
    const int oi = __shfl_xor(best_ss, off);
    if (ov < best || (ov == best && oi < best_ss)) { best = ov; best_ss = oi; }
  }
#ifdef PH
  PH(5);
#endif
  ok = 0;
  bits40 = 0ull;
  if (best < INFINITY) {
    bits40 = vit_retrace_wave(d_est[0], d_est[1], d_est[2], __builtin_amdgcn_readfirstlane(best_ss), lane);
#ifdef PH
    PH(7);
#endif
    ok = pbch_crc_ok(bits40, n_ports);
  }
}
