// sss_foe.hip -- SSS maximum-likelihood detection and PSS/SSS fine frequency-offset estimate.
//
// Replaces extract_psss (ref src/searcher.cpp:516-530), sss_detect_getce_sss (:533-632),
// sss_detect_ml(_helper) (:636-693), sss_detect (:696-761) and pss_sss_foe (:767-850).
//
// All arithmetic in fp64 like the reference.  Per peak the work is tiny and latency-bound, so it
// is spread as widely as the data dependences allow:
//   k_sss_win    numbers the (buffer, peak) pairs of the whole batch (workgroup 0 writes the list for the kernels behind
//                it; k_peak_list does that alone in front of the pss_sss_foe stage entry point);
//                one workgroup per (peak, half-frame occurrence), one wave per 128-sample window
//                (PSS, extended-CP SSS, normal-CP SSS): frequency-correct while staging into LDS,
//                direct 62x128 DFT of the PSS/SSS subcarriers only (<= 60 windows per peak: an FFT
//                would buy nothing and the direct form is the more accurate one), channel
//                smoothing and noise power of the occurrence -> workspace record;
//   k_sss_ml     one workgroup per peak: even/odd combining in the reference's k order, the
//                168 x 2 x 2 ML search and the decision;
//   k_foe_win    one workgroup per (peak, occurrence): PSS and SSS windows, per-occurrence FOE term;
//   k_foe_fin    one thread per peak: sum the terms in occurrence order -> freq_fine.
// Every workgroup needs < 8 KB of LDS (k_sss_ml 17 KB) and at most 4 waves, so they can also be
// placed next to resident correlation workgroups of the following batch.
#include "lcs_internal.h"

#define SF_THREADS 256
#define MAX_HF 20
#define FS_LTE 30720000.0

struct cd2 { double re, im; };
__device__ __forceinline__ cd2 mk(double a, double b) { cd2 r; r.re = a; r.im = b; return r; }
// exp(j x): one sincos call (one argument reduction; cos(x) and sin(x) as two calls cost 1.8 x the instructions)
__device__ __forceinline__ cd2 cis(double x) { double s_, c_; sincos(x, &s_, &c_); return mk(c_, s_); }
__device__ __forceinline__ cd2 cadd(cd2 a, cd2 b) { return mk(a.re + b.re, a.im + b.im); }
__device__ __forceinline__ cd2 csub(cd2 a, cd2 b) { return mk(a.re - b.re, a.im - b.im); }
__device__ __forceinline__ cd2 cmul(cd2 a, cd2 b) { return mk(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
__device__ __forceinline__ cd2 cconj(cd2 a) { return mk(a.re, -a.im); }
__device__ __forceinline__ cd2 cscale(cd2 a, double s) { return mk(a.re * s, a.im * s); }
__device__ __forceinline__ cd2 cdivr(cd2 a, double s) { return mk(a.re / s, a.im / s); }
__device__ __forceinline__ double cabs2(cd2 a) { return a.re * a.re + a.im * a.im; }

__device__ __forceinline__ int d_round_i(double x) { return (int)rint(x); }
__device__ __forceinline__ int d_floor_i(double x) { return (int)floor(x); }
__device__ __forceinline__ double d_matlab_mod(double k, double n) { return (n == 0) ? k : (k - n * d_floor_i(k / n)); }
__device__ __forceinline__ double d_wrap(double x, double sm, double lg) { return d_matlab_mod(x - sm, lg - sm) + sm; }
__device__ __forceinline__ int d_range_len(double first, double incr, double last) {   // ref src/itpp_ext.cpp:97-109
  const double s1 = (double)((last - first > 0) - (last - first < 0));
  const double s2 = (double)((incr > 0) - (incr < 0));
  return (s1 * s2 >= 0) ? d_floor_i((last - first) / incr) + 1 : 0;
}

// workspace record of one (peak, occurrence), in doubles
#define SW_HSM 0       // 62 complex: smoothed PSS channel
#define SW_NRM 124     // 62 complex: normal-CP SSS bins
#define SW_EXT 248     // 62 complex: extended-CP SSS bins
#define SW_NP 372      // noise power of the occurrence
#define SW_ACC 374     // FOE: complex per-occurrence term
#define SW_REC 376
#define SW_ITEM ((size_t)MAX_HF * SW_REC)

// Twiddles exp(-j 2 pi m / 128) for the workgroup.
__device__ __forceinline__ void fill_twiddles(cd2 *W, int tid) {
  if (tid < 128) { double s, c; sincospi((double)tid / 64.0, &s, &c); W[tid] = mk(c, -s); }
}

// capbuf.mid(loc,128) -> fshift(., foc_freq, fs) -> rotate left by 2 (ref :523-525), one sample
__device__ __forceinline__ cd2 stage_sample(const CapView &cap, uint32_t n_cap, long loc, double k, int n) {
  const int t = (n + 2) & 127;
  const long src = loc + t;
  cd2 v = mk(0, 0);
  if (src >= 0 && (uint64_t)src < n_cap) { const double2 c = cap_at(cap, (size_t)src); v = mk(c.x, c.y); }
  double sn, cs;
  sincos(k * (double)t, &sn, &cs);
  return cmul(v, mk(cs, sn));
}

// 128-point transform of one staged window by ONE wave, in place (decimation in frequency: natural order in,
// bit-reversed order out), one butterfly per lane and stage; tw[stg] = this lane's twiddle of stage stg, held in registers
// (fft_twiddles).  Rounds 1-3 summed each of the 62 wanted bins directly (128 terms per bin, twiddles gathered from LDS at
// strides that collide on the banks): 10 x the operations of the 7 stages.  No other wave touches the window and a wave's
// LDS accesses execute in program order, so no workgroup barrier is needed between the stages.
__device__ __forceinline__ void fft_twiddles(const cd2 *W, int lane, cd2 (&tw)[7]) {
#pragma unroll
  for (int stg = 0; stg < 7; ++stg) tw[stg] = W[(lane & ((64 >> stg) - 1)) << stg];
}
__device__ __forceinline__ void fft128_wave(cd2 *x, const cd2 (&tw)[7], int lane) {
#pragma unroll
  for (int stg = 0; stg < 7; ++stg) {
    const int half = 64 >> stg;
    const int pos = lane & (half - 1);
    const int i0 = ((lane >> (6 - stg)) << (7 - stg)) + pos, i1 = i0 + half;
    const cd2 a = x[i0], b = x[i1];
    x[i0] = cadd(a, b);
    x[i1] = cmul(csub(a, b), tw[stg]);
    lcs_wave_sync();
  }
}
// One of the 62 PSS/SSS bins [97..127, 1..31] of a transformed window, /sqrt(128) (ref :527-529)
__device__ __forceinline__ cd2 fft62_bin(const cd2 *x, int bin_idx) {
  const int bin = (bin_idx < 31) ? 97 + bin_idx : bin_idx - 30;
  return cdivr(x[__brev((unsigned)bin) >> 25], sqrt(128.0));
}

// h_raw -> h_sm (13-tap mean, ref :584-588) for subcarrier t
__device__ __forceinline__ cd2 smooth13(const cd2 *h_raw, int t) {
  const int lt = (t - 6 > 0) ? t - 6 : 0, rt = (t + 6 < 61) ? t + 6 : 61;
  cd2 s = mk(0, 0);
  for (int i = lt; i <= rt; ++i) s = cadd(s, h_raw[i]);
  return cdivr(s, (double)(rt - lt + 1));
}
// sigpower(h_sm - h_raw) (ref :591), subcarrier order
__device__ __forceinline__ double noise_power(const cd2 *h_sm, const cd2 *h_raw) {
  double r = 0;
  for (int t = 0; t < 62; ++t) { const cd2 d = csub(h_sm[t], h_raw[t]); r += d.re * d.re + d.im * d.im; }
  return r / 62;
}

// ------------------------------------------------------------------ work list of peaks
// The peaks of a batch, numbered in (buffer, peak) order.  One wave: lane = capture buffer, 64 at a time.
__device__ __forceinline__ int peak_count(const int *__restrict__ npeaks, int n_buf, int s) {
  return (s < n_buf) ? min(max(npeaks[s], 0), LCS_MAXP) : 0;
}
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
  for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(v, off); if (lane >= off) v += u; }
  return v;
}
__device__ void peak_list_write(const int *__restrict__ npeaks, int n_buf, WorkItem *__restrict__ items, int *__restrict__ n_items, int lane) {
  int base = 0;
  for (int s0 = 0; s0 < n_buf; s0 += 64) {
    const int s = s0 + lane;
    const int cnt = peak_count(npeaks, n_buf, s);
    const int incl = wave_incl_scan(cnt, lane);
    const int at = base + incl - cnt;
    for (int p = 0; p < cnt; ++p) { items[at + p].slot = s; items[at + p].peak = p; }
    base += __shfl(incl, 63);
  }
  if (lane == 0) *n_items = base;
}
// the stage entry point of pss_sss_foe alone (no k_sss_win in front of it)
__global__ __launch_bounds__(64) void k_peak_list(const int *__restrict__ npeaks, int n_buf, WorkItem *__restrict__ items,
                                                  int *__restrict__ n_items) {
  LCS_TAIL_PRIO();
  peak_list_write(npeaks, n_buf, items, n_items, threadIdx.x);
}
// The same numbering without a list, for the kernel that runs right behind the peak search (round 2: a one-wave
// kernel in between): every wave calls these with all 64 lanes; results are wave-uniform.
__device__ __forceinline__ int peak_total(const int *__restrict__ npeaks, int n_buf, int lane) {
  int base = 0;
  for (int s0 = 0; s0 < n_buf; s0 += 64) base += __shfl(wave_incl_scan(peak_count(npeaks, n_buf, s0 + lane), lane), 63);
  return base;
}
__device__ __forceinline__ WorkItem peak_lookup(const int *__restrict__ npeaks, int n_buf, int it, int lane) {
  WorkItem w;
  w.slot = 0; w.peak = 0;
  int base = 0;
  for (int s0 = 0; s0 < n_buf; s0 += 64) {
    const int cnt = peak_count(npeaks, n_buf, s0 + lane);
    const int incl = wave_incl_scan(cnt, lane);
    const int total = __shfl(incl, 63);
    if (it < base + total) {                                            // wave-uniform
      const unsigned long long m = __ballot(it < base + incl);          // first lane whose range reaches past `it`
      const int src = __ffsll((long long)m) - 1;
      w.slot = s0 + src;
      w.peak = it - base - __shfl(incl - cnt, src);
      return w;
    }
    base += total;
  }
  return w;
}

// ------------------------------------------------------------------ sss_detect geometry
struct SssGeo { double peak_loc, k_factor, kph; int n_pss; };
__device__ __forceinline__ SssGeo sss_geometry(const lcs_cell &cell, const SlotParams &p, uint32_t n_cap) {
  SssGeo g;
  g.peak_loc = cell.ind;
  g.k_factor = (p.fc_req - cell.freq) / p.fc_prog;
  if (g.peak_loc + 9 < 162) g.peak_loc += 9600 * g.k_factor;
  g.n_pss = d_range_len(g.peak_loc, g.k_factor * 9600, (double)n_cap - 125 - 9);
  if (g.n_pss > MAX_HF) g.n_pss = MAX_HF;
  const double fs = p.fs_prog * g.k_factor;
  g.kph = M_PI * (-cell.freq) / (fs / 2);
  return g;
}

#define SW_THREADS 192
__global__ __launch_bounds__(SW_THREADS) void k_sss_win(const lcs_cell *__restrict__ peaks, const int *__restrict__ npeaks, int n_buf,
                                                        WorkItem *__restrict__ items, int *__restrict__ n_items,
                                                        const CapSrc src,
                                                        uint32_t n_cap, const SlotParams *__restrict__ params,
                                                        const double2 *__restrict__ pss_fd, double *__restrict__ ws) {
  LCS_TAIL_PRIO();
  __shared__ cd2 W[128];
  __shared__ cd2 win[3][128];
  __shared__ cd2 h_raw[62], h_sm[62];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  fill_twiddles(W, tid);
  __syncthreads();
  cd2 tw[7];
  fft_twiddles(W, lane, tw);
  // the work list: numbered here from the per-buffer counts; workgroup 0 also writes it out for the kernels that follow
  if (blockIdx.x == 0 && w == 0) peak_list_write(npeaks, n_buf, items, n_items, lane);
  const int n_jobs = peak_total(npeaks, n_buf, lane) * MAX_HF;
  for (int job = blockIdx.x; job < n_jobs; job += gridDim.x) {
    const int it = job / MAX_HF, k = job % MAX_HF;
    const WorkItem wi = peak_lookup(npeaks, n_buf, it, lane);
    const int slot = wi.slot;
    const lcs_cell cell = peaks[(size_t)slot * LCS_MAXP + wi.peak];
    const SlotParams p = params[slot];
    const SssGeo g = sss_geometry(cell, p, n_cap);
    if (k >= g.n_pss) continue;
    const CapView cap = cap_view(src, slot);
    double *rec = ws + (size_t)it * SW_ITEM + (size_t)k * SW_REC;
    __syncthreads();
    {   // the PSS window, the extended-CP SSS window and the normal-CP SSS window (ref :578-597)
      const uint32_t pss_loc = (uint32_t)d_round_i(g.peak_loc + k * (g.k_factor * 9600));
      const long pss_dft = (long)(pss_loc + 9 - 2);
      const long loc = (w == 0) ? pss_dft : (w == 1 ? pss_dft - 128 - 32 : pss_dft - 128 - 9);
      win[w][lane] = stage_sample(cap, n_cap, loc, g.kph, lane);
      win[w][lane + 64] = stage_sample(cap, n_cap, loc, g.kph, lane + 64);
    }
    lcs_wave_sync();          // wave w staged window w itself
    fft128_wave(win[w], tw, lane);
    if (lane < 62) {
      const cd2 o = fft62_bin(win[w], lane);
      if (w == 0) { const double2 f = pss_fd[cell.n_id_2 * 62 + lane]; h_raw[lane] = cmul(o, mk(f.x, -f.y)); }
      else { rec[(w == 1 ? SW_EXT : SW_NRM) + 2 * lane] = o.re; rec[(w == 1 ? SW_EXT : SW_NRM) + 2 * lane + 1] = o.im; }
    }
    __syncthreads();
    if (tid < 62) {
      const cd2 v = smooth13(h_raw, tid);
      h_sm[tid] = v;
      rec[SW_HSM + 2 * tid] = v.re; rec[SW_HSM + 2 * tid + 1] = v.im;
    }
    __syncthreads();
    if (tid == 0) rec[SW_NP] = noise_power(h_sm, h_raw);
  }
}

// ------------------------------------------------------------------ combining, ML, decision
struct MlShared {
  double np12[124];
  double rnp12[124];          // 1/np12
  cd2 nrm12[124];
  cd2 ext12[124];
  double ll[2][2][168];       // [nrm/ext][column][n_id_1]
  double dec[4][4];
};

__global__ __launch_bounds__(SF_THREADS) void k_sss_ml(lcs_cell *__restrict__ peaks, const WorkItem *__restrict__ items,
                                                       const int *__restrict__ n_items, uint32_t n_cap,
                                                       const SlotParams *__restrict__ params, double thresh2,
                                                       const int8_t *__restrict__ sss_fd, const double *__restrict__ ws,
                                                       double *dbg) {
  LCS_TAIL_PRIO();
  __shared__ MlShared S;
  const int tid = threadIdx.x;
  for (int it = blockIdx.x; it < *n_items; it += gridDim.x) {
    const int slot = items[it].slot;
    lcs_cell *cell_p = peaks + (size_t)slot * LCS_MAXP + items[it].peak;
    const lcs_cell cell = *cell_p;
    const SlotParams p = params[slot];
    const SssGeo g = sss_geometry(cell, p, n_cap);
    if (g.n_pss < 1) continue;
    const int n_id_2 = cell.n_id_2;
    const double *wsi = ws + (size_t)it * SW_ITEM;
    __syncthreads();
    // combine even (h1) / odd (h2) occurrences per subcarrier (ref :618-631)
    if (tid < 124) {
      const int h = tid / 62, t = tid % 62;
      double s = 0;
      cd2 sn = mk(0, 0), se = mk(0, 0);
      for (int k = h; k < g.n_pss; k += 2) {
        const double *rec = wsi + (size_t)k * SW_REC;
        const cd2 hs = mk(rec[SW_HSM + 2 * t], rec[SW_HSM + 2 * t + 1]);
        const double rnp = 1.0 / rec[SW_NP];
        s += cabs2(hs) * rnp;
        const cd2 w = cmul(cconj(hs), mk(rnp, 0));
        sn = cadd(sn, cmul(w, mk(rec[SW_NRM + 2 * t], rec[SW_NRM + 2 * t + 1])));
        se = cadd(se, cmul(w, mk(rec[SW_EXT + 2 * t], rec[SW_EXT + 2 * t + 1])));
      }
      const double np_est = 1 / (1 + s);
      S.np12[tid] = np_est;
      S.rnp12[tid] = 1.0 / np_est;
      S.nrm12[tid] = cscale(sn, np_est);
      S.ext12[tid] = cscale(se, np_est);
    }
    __syncthreads();
    // ML over 168 n_id_1 x {12,21} x {nrm,ext} (ref :636-693)
    for (int job = tid; job < 168 * 4; job += SF_THREADS) {
      const int n1 = job >> 2, col = job & 1, ext = (job >> 1) & 1;
      const cd2 *est = ext ? S.ext12 : S.nrm12;
      const int8_t *h1 = sss_fd + ((n1 * 3 + n_id_2) * 2 + 0) * 62;
      const int8_t *h2 = sss_fd + ((n1 * 3 + n_id_2) * 2 + 1) * 62;
      const int8_t *first = col ? h2 : h1, *second = col ? h1 : h2;
      cd2 acc = mk(0, 0);
      for (int i = 0; i < 124; ++i) {
        const double tv = (double)(i < 62 ? first[i] : second[i - 62]);
        acc = cadd(acc, cmul(cconj(est[i]), mk(tv, 0)));
      }
      const double ang = atan2(acc.im, acc.re);
      const cd2 rot = cis(-ang);
      double s1 = 0, s2 = 0;
      for (int i = 0; i < 124; ++i) {      // the two sums of ref :649 keep their own order; x/np as x*(1/np)
        const double tv = (double)(i < 62 ? first[i] : second[i - 62]);
        const cd2 d = csub(cmul(mk(tv, 0), rot), est[i]);
        s1 += (d.re * d.re) * S.rnp12[i];
        s2 += (d.im * d.im) * S.rnp12[i];
      }
      S.ll[ext][col][n1] = -s1 - s2;
    }
    __syncthreads();
    // decision (ref :719-758): lanes 0..3 each scan one of the four likelihood columns (max, first
    // arg-max, sum, sum of squares in index order); lane 0 combines them in the reference's order
    if (tid < 4) {
      const double *col = &S.ll[tid >> 1][tid & 1][0];
      double mx = col[0], sum = 0, sq = 0;
      int am = 0;
      for (int t = 0; t < 168; ++t) {
        const double v = col[t];
        if (v > mx) { mx = v; am = t; }
        sum += v; sq += v * v;
      }
      S.dec[tid][0] = mx; S.dec[tid][1] = (double)am; S.dec[tid][2] = sum; S.dec[tid][3] = sq;
    }
    __syncthreads();
    if (tid == 0) {
      const double mx_n = (S.dec[1][0] > S.dec[0][0]) ? S.dec[1][0] : S.dec[0][0];
      const double mx_e = (S.dec[3][0] > S.dec[2][0]) ? S.dec[3][0] : S.dec[2][0];
      const int e = (mx_n > mx_e) ? 0 : 1;
      const int cp_type = e ? LCS_CP_EXTENDED : LCS_CP_NORMAL;
      const double mx0 = S.dec[2 * e][0], mx1 = S.dec[2 * e + 1][0];
      const double k_factor = g.k_factor;
      double frame_start = cell.ind + (128 + 9 - 960 - 2) * 16 / FS_LTE * p.fs_prog * k_factor;
      int col;
      if (mx0 > mx1) col = 0;
      else { col = 1; frame_start = frame_start + 9600 * k_factor * 16 / FS_LTE * p.fs_prog * k_factor; }   // k_factor^2: quirk Q3
      frame_start = d_wrap(frame_start, -0.5, (2 * 9600.0 - 0.5) * 16 / FS_LTE * p.fs_prog * k_factor);
      const int n_id_1_est = (int)S.dec[2 * e + col][1];
      const double lik_final = S.dec[2 * e + col][0];
      double sum = 0, sq = 0;
      for (int q = 0; q < 4; ++q) { sum += S.dec[q][2]; sq += S.dec[q][3]; }
      const int len = 672;
      const double lik_mean = sum / len;
      const double lik_var = (sq - sum * sum / len) / (len - 1);    // itpp::variance (unbiased)
      if (lik_final >= lik_mean + sqrt(lik_var) * thresh2) {
        cell_p->n_id_1 = n_id_1_est;
        cell_p->cp_type = cp_type;
        cell_p->frame_start = frame_start;
      }
    }
    if (dbg) {   // the reference's "only used for testing" outputs
      for (int i = tid; i < 62; i += SF_THREADS) {
        dbg[i] = S.np12[i]; dbg[62 + i] = S.np12[62 + i];
        dbg[124 + 2 * i] = S.nrm12[i].re; dbg[124 + 2 * i + 1] = S.nrm12[i].im;
        dbg[248 + 2 * i] = S.nrm12[62 + i].re; dbg[248 + 2 * i + 1] = S.nrm12[62 + i].im;
        dbg[372 + 2 * i] = S.ext12[i].re; dbg[372 + 2 * i + 1] = S.ext12[i].im;
        dbg[496 + 2 * i] = S.ext12[62 + i].re; dbg[496 + 2 * i + 1] = S.ext12[62 + i].im;
      }
      for (int i = tid; i < 168 * 2; i += SF_THREADS) {
        dbg[620 + i] = S.ll[0][i & 1][i >> 1];           // log_lik_nrm [168][2]
        dbg[620 + 336 + i] = S.ll[1][i & 1][i >> 1];     // log_lik_ext [168][2]
      }
    }
  }
}

// ------------------------------------------------------------------ pss_sss_foe
struct FoeGeo { int pss_sss_dist, sn_init, n_sss; double first_sss, step, k_factor, kph; bool ok; };
__device__ __forceinline__ FoeGeo foe_geometry(const lcs_cell &cell, const SlotParams &p, uint32_t n_cap) {
  FoeGeo g;
  g.ok = false;
  g.n_sss = 0;
  if (cell.n_id_1 < 0) return g;
  const double k_factor = (p.fc_req - cell.freq) / p.fc_prog;
  g.k_factor = k_factor;
  if (cell.cp_type == LCS_CP_NORMAL) {
    g.pss_sss_dist = (int)(uint16_t)d_round_i((128 + 9) * 16 / FS_LTE * p.fs_prog * k_factor);
    g.first_sss = cell.frame_start + (960 - 128 - 9 - 128) * 16 / FS_LTE * p.fs_prog * k_factor;
  } else if (cell.cp_type == LCS_CP_EXTENDED) {
    g.pss_sss_dist = (int)(uint16_t)d_round_i((128 + 32) * k_factor);   // quirk Q4
    g.first_sss = cell.frame_start + (960 - 128 - 32 - 128) * 16 / FS_LTE * p.fs_prog * k_factor;
  } else return g;
  g.first_sss = d_wrap(g.first_sss, -0.5, 9600 * 2 - 0.5);
  if (g.first_sss - 9600 * k_factor > -0.5) { g.first_sss -= 9600 * k_factor; g.sn_init = 10; } else g.sn_init = 0;
  g.step = 9600 * 16 / FS_LTE * p.fs_prog * k_factor;
  g.n_sss = d_range_len(g.first_sss, g.step, (double)((int)n_cap - 127 - g.pss_sss_dist - 100));
  if (g.n_sss > MAX_HF) g.n_sss = MAX_HF;
  const double fs = p.fs_prog * k_factor;
  g.kph = M_PI * (-cell.freq) / (fs / 2);
  g.ok = true;
  return g;
}

#define FW_THREADS 128
__global__ __launch_bounds__(FW_THREADS) void k_foe_win(const lcs_cell *__restrict__ peaks, const WorkItem *__restrict__ items,
                                                        const int *__restrict__ n_items,
                                                        const CapSrc src,
                                                        uint32_t n_cap, const SlotParams *__restrict__ params,
                                                        const double2 *__restrict__ pss_fd, const int8_t *__restrict__ sss_fd,
                                                        double *__restrict__ ws) {
  LCS_TAIL_PRIO();
  __shared__ cd2 W[128];
  __shared__ cd2 win[2][128];
  __shared__ cd2 h_raw[62], h_sm[62], aux[62];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  fill_twiddles(W, tid);
  __syncthreads();
  cd2 tw[7];
  fft_twiddles(W, lane, tw);
  const int n_jobs = *n_items * MAX_HF;
  for (int job = blockIdx.x; job < n_jobs; job += gridDim.x) {
    const int it = job / MAX_HF, k = job % MAX_HF;
    const int slot = items[it].slot;
    const lcs_cell cell = peaks[(size_t)slot * LCS_MAXP + items[it].peak];
    const SlotParams p = params[slot];
    const FoeGeo g = foe_geometry(cell, p, n_cap);
    if (!g.ok || k >= g.n_sss) continue;
    const CapView cap = cap_view(src, slot);
    double *rec = ws + (size_t)it * SW_ITEM + (size_t)k * SW_REC;
    __syncthreads();
    {
      const uint32_t sss_loc = (uint32_t)d_round_i(g.first_sss + k * g.step);
      const long loc = (w == 0) ? (long)(sss_loc + g.pss_sss_dist) : (long)sss_loc;
      win[w][lane] = stage_sample(cap, n_cap, loc, g.kph, lane);
      win[w][lane + 64] = stage_sample(cap, n_cap, loc, g.kph, lane + 64);
    }
    lcs_wave_sync();          // wave w staged window w itself
    fft128_wave(win[w], tw, lane);
    if (lane < 62) {
      const cd2 o = fft62_bin(win[w], lane);
      if (w == 0) { const double2 f = pss_fd[cell.n_id_2 * 62 + lane]; h_raw[lane] = cmul(o, mk(f.x, -f.y)); }
      else {
        // exp(J*pi*-freq/(FS_LTE/16/2)*-pss_sss_dist), evaluated left to right (ref :832)
        double ph_im = M_PI;
        ph_im = ph_im * (-cell.freq);
        ph_im = ph_im / (FS_LTE / 16 / 2);
        ph_im = ph_im * (double)(-g.pss_sss_dist);
        const cd2 ph = cis(ph_im);
        // the slot number toggles with every occurrence, starting from sn_init (ref :800, :813)
        const int sn = ((k & 1) == 0) ? g.sn_init : 10 - g.sn_init;
        const double sf = (double)sss_fd[((cell.n_id_1 * 3 + cell.n_id_2) * 2 + (sn != 0)) * 62 + lane];
        aux[lane] = cmul(cmul(o, ph), mk(sf, 0));
      }
    }
    __syncthreads();
    if (tid < 62) h_sm[tid] = smooth13(h_raw, tid);
    __syncthreads();
    if (tid == 0) {     // sum over the 62 subcarriers, in subcarrier order (ref :836-843)
      const double np = noise_power(h_sm, h_raw);
      cd2 acc = mk(0, 0);
      for (int t = 0; t < 62; ++t) {
        const double a2 = cabs2(h_sm[t]);
        const double wgt = a2 * (1.0 / (2 * a2 * np + np * np));
        acc = cadd(acc, cmul(cmul(cconj(aux[t]), h_raw[t]), mk(wgt, 0)));
      }
      rec[SW_ACC] = acc.re; rec[SW_ACC + 1] = acc.im;
    }
  }
}

__global__ __launch_bounds__(64) void k_foe_fin(lcs_cell *__restrict__ peaks, const WorkItem *__restrict__ items,
                                                const int *__restrict__ n_items, uint32_t n_cap,
                                                const SlotParams *__restrict__ params, const double *__restrict__ ws) {
  LCS_TAIL_PRIO();
  const int it = blockIdx.x * blockDim.x + threadIdx.x;
  if (it >= *n_items) return;
  const int slot = items[it].slot;
  lcs_cell *cell_p = peaks + (size_t)slot * LCS_MAXP + items[it].peak;
  const lcs_cell cell = *cell_p;
  const SlotParams p = params[slot];
  const FoeGeo g = foe_geometry(cell, p, n_cap);
  if (!g.ok) return;
  cd2 M = mk(0, 0);
  for (int k = 0; k < g.n_sss; ++k) {
    const double *rec = ws + (size_t)it * SW_ITEM + (size_t)k * SW_REC;
    M = cadd(M, mk(rec[SW_ACC], rec[SW_ACC + 1]));
  }
  cell_p->freq_fine = cell.freq + atan2(M.im, M.re) / (2 * M_PI) / (1 / (p.fs_prog * g.k_factor) * g.pss_sss_dist);
}

// ------------------------------------------------------------------ launchers
// mode bit 0: run sss_detect, bit 1: run pss_sss_foe (only for cells whose SSS was found)
static int run_sss_foe(lcs_ctx *c, int n_buf, uint32_t n_cap, double thresh2, int mode, double *dbg) {
  const size_t cap_items = (size_t)n_buf * LCS_MAXP;
  if (cap_items > c->sss_ws_items) {
    if (c->sss_ws) { HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->sss_ws); c->sss_ws = nullptr; }
    if (c->pk_items) { (void)hipFree(c->pk_items); c->pk_items = nullptr; }
    HIPCHK(c, hipMalloc((void **)&c->sss_ws, cap_items * SW_ITEM * sizeof(double)));
    HIPCHK(c, hipMalloc((void **)&c->pk_items, cap_items * sizeof(WorkItem)));
    if (!c->n_pk) HIPCHK(c, hipMalloc((void **)&c->n_pk, 4 * sizeof(int)));
    c->sss_ws_items = cap_items;
  }
  const CapSrc src = lcs_cap_src(c, n_cap);
  // enough workgroups for every (peak, occurrence) of a typical batch to be resident at once; the
  // kernels loop over the work list, so larger batches only take more rounds
  const int win_grid = (int)std::min<size_t>(cap_items * MAX_HF, LCS_WIN_GRID);
  const int item_grid = (int)std::min<size_t>(cap_items, LCS_ITEM_GRID);
  if (!(mode & 1)) hipLaunchKernelGGL(k_peak_list, dim3(1), dim3(64), 0, c->stream, c->npeaks, n_buf, c->pk_items, c->n_pk);
  if (mode & 1) {
    hipLaunchKernelGGL(k_sss_win, dim3(win_grid), dim3(SW_THREADS), 0, c->stream, c->peaks, c->npeaks, n_buf, c->pk_items, c->n_pk, src,
                       n_cap, c->params, c->d_pss_fd, c->sss_ws);
    hipLaunchKernelGGL(k_sss_ml, dim3(item_grid), dim3(SF_THREADS), 0, c->stream, c->peaks, c->pk_items, c->n_pk, n_cap,
                       c->params, thresh2, c->d_sss_fd, c->sss_ws, dbg);
  }
  if (mode & 2) {
    hipLaunchKernelGGL(k_foe_win, dim3(win_grid), dim3(FW_THREADS), 0, c->stream, c->peaks, c->pk_items, c->n_pk, src,
                       n_cap, c->params, c->d_pss_fd, c->d_sss_fd, c->sss_ws);
    hipLaunchKernelGGL(k_foe_fin, dim3((unsigned)((cap_items + 63) / 64)), dim3(64), 0, c->stream, c->peaks, c->pk_items,
                       c->n_pk, n_cap, c->params, c->sss_ws);
  }
  HIPCHK(c, hipGetLastError());
  return LCS_OK;
}

int lcs_launch_sss_foe(lcs_ctx *c, int n_buf, uint32_t n_cap, double thresh2_n_sigma, double *dbg) {
  return run_sss_foe(c, n_buf, n_cap, thresh2_n_sigma, 3, dbg);
}
// Single-cell helpers for the stage entry points: peaks[0] of slot 0 holds the cell (npeaks[0] = 1).
int lcs_launch_sss_only(lcs_ctx *c, uint32_t n_cap, double thresh2_n_sigma, double *dbg) {
  return run_sss_foe(c, 1, n_cap, thresh2_n_sigma, 1, dbg);
}
int lcs_launch_foe_only(lcs_ctx *c, uint32_t n_cap) { return run_sss_foe(c, 1, n_cap, 0.0, 2, nullptr); }
