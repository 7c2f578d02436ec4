// sss_foe.hip -- SSS maximum-likelihood detection and PSS/SSS fine frequency-offset estimate.
//
// Replaces extract_psss (ref src/searcher.cpp:516-530), sss_detect_getce_sss (:533-632),
// sss_detect_ml(_helper) (:636-693), sss_detect (:696-761) and pss_sss_foe (:767-850).
//
// One 256-thread workgroup per detected PSS peak; all arithmetic in fp64 like the reference.
// Every 128-sample window is frequency-corrected while it is staged into LDS and only the 62
// PSS/SSS subcarriers are evaluated (a direct 62x128 DFT per window: <= 48 windows per peak,
// so an FFT would buy nothing and the direct form is the more accurate one).  Per peak the
// work is tiny and latency-bound; parallelism comes from batching all peaks of all capture
// buffers of a sweep into one launch (grid = peaks x buffers).
#include "lcs_internal.h"

#define SF_THREADS 256
#define MAX_HF 20
#define FS_LTE 30720000.0

struct cd2 { double re, im; };
__device__ __forceinline__ cd2 mk(double a, double b) { cd2 r; r.re = a; r.im = b; return r; }
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

// LDS image of one workgroup.  Occurrences (half-frames) are processed PASS_OCC at a time:
// all their windows are staged, transformed, smoothed in parallel.
#define PASS_OCC 8
#define PASS_WIN (PASS_OCC * 3)
struct SfShared {
  cd2 W[128];                 // exp(-j 2 pi m / 128)
  cd2 win[PASS_WIN][128];     // frequency-corrected, 2-sample-rotated DFT inputs of one pass
  cd2 h_raw[PASS_OCC][62];
  cd2 aux[PASS_OCC][62];      // FOE: SSS bins of the pass
  cd2 h_sm[MAX_HF][62];
  cd2 s_nrm[MAX_HF][62];
  cd2 s_ext[MAX_HF][62];
  double pss_np[MAX_HF];
  double np12[124];
  double rnp12[124];          // 1/np12
  cd2 nrm12[124];
  cd2 ext12[124];
  double ll[2][2][168];       // [nrm/ext][column][n_id_1]
  cd2 acc_k[MAX_HF];
  double dec[4][4];
};

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

// DFT of n_win staged windows at the 62 PSS/SSS bins [97..127, 1..31], /sqrt(128) (ref :527-529).
// Thread (bin = tid % 62, lane group = tid / 62) handles windows group, group+4, ...: the twiddle
// is read once per sample and reused for every window of the thread.  Sums run over n ascending.
template <int MAXW>
__device__ __forceinline__ void dft62_multi(const SfShared &S, int n_win, int tid, cd2 *out /*[MAXW]*/, int &bin_idx, int &grp) {
  bin_idx = tid % 62;
  grp = tid / 62;
  const int bin = (bin_idx < 31) ? 97 + bin_idx : bin_idx - 30;
#pragma unroll
  for (int i = 0; i < MAXW; ++i) out[i] = mk(0, 0);
  if (tid >= 248) return;
  for (int n = 0; n < 128; ++n) {
    const cd2 tw = S.W[(bin * n) & 127];
#pragma unroll
    for (int i = 0; i < MAXW; ++i) {
      const int w = grp + 4 * i;
      if (w < n_win) out[i] = cadd(out[i], cmul(S.win[w][n], tw));
    }
  }
  const double sq = sqrt(128.0);
#pragma unroll
  for (int i = 0; i < MAXW; ++i) out[i] = cdivr(out[i], sq);
}

// h_raw[kk] -> h_sm (13-tap mean, ref :584-588) and pss_np = sigpower(h_sm-h_raw) (ref :591) for
// the nk occurrences of a pass; h_sm rows start at h_sm0.
__device__ void smooth_and_np(SfShared &S, int nk, cd2 (*h_sm0)[62], double *np0, int tid) {
  for (int e = tid; e < nk * 62; e += SF_THREADS) {
    const int kk = e / 62, t = e % 62;
    const int lt = (t - 6 > 0) ? t - 6 : 0, rt = (t + 6 < 61) ? t + 6 : 61;
    cd2 s = mk(0, 0);
    for (int i = lt; i <= rt; ++i) s = cadd(s, S.h_raw[kk][i]);
    h_sm0[kk][t] = cdivr(s, (double)(rt - lt + 1));
  }
  __syncthreads();
  if (tid < nk) {
    double r = 0;
    for (int t = 0; t < 62; ++t) { const cd2 d = csub(h_sm0[tid][t], S.h_raw[tid][t]); r += d.re * d.re + d.im * d.im; }
    np0[tid] = r / 62;
  }
  __syncthreads();
}

__device__ void dev_sss_detect(SfShared &S, lcs_cell &cell, const CapView &cap, uint32_t n_cap,
                               const SlotParams &p, double thresh2, const double2 *__restrict__ pss_fd,
                               const int8_t *__restrict__ sss_fd, double *dbg) {
  const int tid = threadIdx.x;
  double peak_loc = cell.ind;
  const double peak_freq = cell.freq;
  const int n_id_2 = cell.n_id_2;
  const double k_factor = (p.fc_req - peak_freq) / p.fc_prog;
  if (peak_loc + 9 < 162) peak_loc += 9600 * k_factor;
  int n_pss = d_range_len(peak_loc, k_factor * 9600, (double)n_cap - 125 - 9);
  if (n_pss > MAX_HF) n_pss = MAX_HF;
  if (n_pss < 1) return;
  const double fs = p.fs_prog * k_factor;
  const double kph = M_PI * (-peak_freq) / (fs / 2);

  for (int k0 = 0; k0 < n_pss; k0 += PASS_OCC) {
    const int nk = min(PASS_OCC, n_pss - k0);
    // stage the PSS window, the extended-CP SSS window and the normal-CP SSS window of nk occurrences
    for (int e = tid; e < nk * 3 * 128; e += SF_THREADS) {
      const int kk = e / 384, w = (e >> 7) % 3, n = e & 127;
      const uint32_t pss_loc = (uint32_t)d_round_i(peak_loc + (k0 + kk) * (k_factor * 9600));
      const long pss_dft = (long)(pss_loc + 9 - 2);
      const long loc = (w == 0) ? pss_dft : (w == 1 ? pss_dft - 128 - 32 : pss_dft - 128 - 9);
      S.win[kk * 3 + w][n] = stage_sample(cap, n_cap, loc, kph, n);
    }
    __syncthreads();
    cd2 o[6];
    int b, grp;
    dft62_multi<6>(S, nk * 3, tid, o, b, grp);
    if (tid < 248) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int wi = grp + 4 * i;
        if (wi < nk * 3) {
          const int kk = wi / 3, w = wi % 3;
          if (w == 0) { const double2 f = pss_fd[n_id_2 * 62 + b]; S.h_raw[kk][b] = cmul(o[i], mk(f.x, -f.y)); }
          else if (w == 1) S.s_ext[k0 + kk][b] = o[i];
          else S.s_nrm[k0 + kk][b] = o[i];
        }
      }
    }
    __syncthreads();
    smooth_and_np(S, nk, &S.h_sm[k0], &S.pss_np[k0], tid);
  }
  // combine even (h1) / odd (h2) occurrences per subcarrier (ref :618-631)
  if (tid < 124) {
    const int h = tid / 62, t = tid % 62;
    double s = 0;
    for (int k = h; k < n_pss; k += 2) s += cabs2(S.h_sm[k][t]) * (1.0 / S.pss_np[k]);
    const double np_est = 1 / (1 + s);
    cd2 sn = mk(0, 0), se = mk(0, 0);
    for (int k = h; k < n_pss; k += 2) {
      const cd2 w = cmul(cconj(S.h_sm[k][t]), mk(1.0 / S.pss_np[k], 0));
      sn = cadd(sn, cmul(w, S.s_nrm[k][t]));
      se = cadd(se, cmul(w, S.s_ext[k][t]));
    }
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
    const cd2 rot = mk(cos(-ang), sin(-ang));
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
      cell.n_id_1 = n_id_1_est;
      cell.cp_type = cp_type;
      cell.frame_start = frame_start;
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
  __syncthreads();
}

__device__ void dev_pss_sss_foe(SfShared &S, lcs_cell &cell, const CapView &cap, uint32_t n_cap,
                                const SlotParams &p, const double2 *__restrict__ pss_fd,
                                const int8_t *__restrict__ sss_fd) {
  const int tid = threadIdx.x;
  const double k_factor = (p.fc_req - cell.freq) / p.fc_prog;
  int pss_sss_dist;
  double first_sss;
  if (cell.cp_type == LCS_CP_NORMAL) {
    pss_sss_dist = (int)(uint16_t)d_round_i((128 + 9) * 16 / FS_LTE * p.fs_prog * k_factor);
    first_sss = cell.frame_start + (960 - 128 - 9 - 128) * 16 / FS_LTE * p.fs_prog * k_factor;
  } else if (cell.cp_type == LCS_CP_EXTENDED) {
    pss_sss_dist = (int)(uint16_t)d_round_i((128 + 32) * k_factor);   // quirk Q4
    first_sss = cell.frame_start + (960 - 128 - 32 - 128) * 16 / FS_LTE * p.fs_prog * k_factor;
  } else return;
  int sn_init;
  first_sss = d_wrap(first_sss, -0.5, 9600 * 2 - 0.5);
  if (first_sss - 9600 * k_factor > -0.5) { first_sss -= 9600 * k_factor; sn_init = 10; } else sn_init = 0;
  const double step = 9600 * 16 / FS_LTE * p.fs_prog * k_factor;
  int n_sss = d_range_len(first_sss, step, (double)((int)n_cap - 127 - pss_sss_dist - 100));
  if (n_sss > MAX_HF) n_sss = MAX_HF;
  const double fs = p.fs_prog * k_factor;
  const double kph = M_PI * (-cell.freq) / (fs / 2);
  // exp(J*pi*-freq/(FS_LTE/16/2)*-pss_sss_dist), evaluated left to right (ref :832)
  double ph_im = M_PI;
  ph_im = ph_im * (-cell.freq);
  ph_im = ph_im / (FS_LTE / 16 / 2);
  ph_im = ph_im * (double)(-pss_sss_dist);
  const cd2 ph = mk(cos(ph_im), sin(ph_im));
  for (int k0 = 0; k0 < n_sss; k0 += PASS_OCC) {
    const int nk = min(PASS_OCC, n_sss - k0);
    for (int e = tid; e < nk * 2 * 128; e += SF_THREADS) {
      const int kk = e >> 8, w = (e >> 7) & 1, n = e & 127;
      const uint32_t sss_loc = (uint32_t)d_round_i(first_sss + (k0 + kk) * step);
      const long loc = (w == 0) ? (long)(sss_loc + pss_sss_dist) : (long)sss_loc;
      S.win[kk * 2 + w][n] = stage_sample(cap, n_cap, loc, kph, n);
    }
    __syncthreads();
    cd2 o[4];
    int b, grp;
    dft62_multi<4>(S, nk * 2, tid, o, b, grp);
    if (tid < 248) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int wi = grp + 4 * i;
        if (wi < nk * 2) {
          const int kk = wi >> 1, w = wi & 1;
          if (w == 0) { const double2 f = pss_fd[cell.n_id_2 * 62 + b]; S.h_raw[kk][b] = cmul(o[i], mk(f.x, -f.y)); }
          else {
            // the slot number toggles with every occurrence, starting from sn_init (ref :800, :813)
            const int sn = (((k0 + kk) & 1) == 0) ? sn_init : 10 - sn_init;
            const double sf = (double)sss_fd[((cell.n_id_1 * 3 + cell.n_id_2) * 2 + (sn != 0)) * 62 + b];
            S.aux[kk][b] = cmul(cmul(o[i], ph), mk(sf, 0));
          }
        }
      }
    }
    __syncthreads();
    smooth_and_np(S, nk, &S.h_sm[k0], &S.pss_np[k0], tid);
    if (tid < nk) {     // per-occurrence sum over the 62 subcarriers, in subcarrier order (ref :836-843)
      const double np = S.pss_np[k0 + tid];
      cd2 acc = mk(0, 0);
      for (int t = 0; t < 62; ++t) {
        const double a2 = cabs2(S.h_sm[k0 + tid][t]);
        const double w = a2 * (1.0 / (2 * a2 * np + np * np));
        acc = cadd(acc, cmul(cmul(cconj(S.aux[tid][t]), S.h_raw[tid][t]), mk(w, 0)));
      }
      S.acc_k[k0 + tid] = acc;
    }
    __syncthreads();
  }
  if (tid == 0) {
    cd2 M = mk(0, 0);
    for (int k = 0; k < n_sss; ++k) M = cadd(M, S.acc_k[k]);
    cell.freq_fine = cell.freq + atan2(M.im, M.re) / (2 * M_PI) / (1 / (p.fs_prog * k_factor) * pss_sss_dist);
  }
}

// mode bit 0: run sss_detect, bit 1: run pss_sss_foe (only for cells whose SSS was found)
__global__ __launch_bounds__(SF_THREADS) void k_sss_foe(lcs_cell *__restrict__ peaks, const int *__restrict__ npeaks,
                                                         const float2 *__restrict__ cap32,
                                                         const double2 *__restrict__ cap64, uint32_t n_cap,
                                                         const SlotParams *__restrict__ params, double thresh2,
                                                         const double2 *__restrict__ pss_fd,
                                                         const int8_t *__restrict__ sss_fd, int mode, double *dbg) {
  const int slot = blockIdx.y, pk = blockIdx.x;
  if (pk >= npeaks[slot] || pk >= LCS_MAXP) return;
  extern __shared__ __attribute__((aligned(16))) char sf_smem[];
  SfShared &S = *reinterpret_cast<SfShared *>(sf_smem);
  __shared__ lcs_cell cell;
  const int tid = threadIdx.x;
  if (tid < 128) { double s, c; sincospi((double)tid / 64.0, &s, &c); S.W[tid] = mk(c, -s); }
  if (tid == 0) cell = peaks[(size_t)slot * LCS_MAXP + pk];
  __syncthreads();
  const CapView cap = cap_view(cap32, cap64, slot, n_cap);
  const SlotParams p = params[slot];
  if (mode & 1) dev_sss_detect(S, cell, cap, n_cap, p, thresh2, pss_fd, sss_fd, dbg);
  __syncthreads();
  if ((mode & 2) && cell.n_id_1 >= 0) dev_pss_sss_foe(S, cell, cap, n_cap, p, pss_fd, sss_fd);
  __syncthreads();
  if (tid == 0) peaks[(size_t)slot * LCS_MAXP + pk] = cell;
}

static int sf_attr(lcs_ctx *c) {
  static bool done = false;
  if (!done) {
    HIPCHK(c, hipFuncSetAttribute((const void *)k_sss_foe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SfShared)));
    done = true;
  }
  return LCS_OK;
}

int lcs_launch_sss_foe(lcs_ctx *c, int n_buf, uint32_t n_cap, double thresh2_n_sigma, double *dbg) {
  int rc = sf_attr(c);
  if (rc) return rc;
  hipLaunchKernelGGL(k_sss_foe, dim3(LCS_MAXP, n_buf), dim3(SF_THREADS), sizeof(SfShared), c->stream, c->peaks, c->npeaks, c->cap32, c->cap64_valid ? c->cap64 : nullptr, n_cap,
                     c->params, thresh2_n_sigma, c->d_pss_fd, c->d_sss_fd, 3, dbg);
  HIPCHK(c, hipGetLastError());
  return LCS_OK;
}

// Single-cell helpers for the stage entry points: peaks[0] of slot 0 holds the cell.
int lcs_launch_sss_only(lcs_ctx *c, uint32_t n_cap, double thresh2_n_sigma, double *dbg) {
  int rc = sf_attr(c);
  if (rc) return rc;
  hipLaunchKernelGGL(k_sss_foe, dim3(1, 1), dim3(SF_THREADS), sizeof(SfShared), c->stream, c->peaks, c->npeaks, c->cap32, c->cap64_valid ? c->cap64 : nullptr, n_cap, c->params,
                     thresh2_n_sigma, c->d_pss_fd, c->d_sss_fd, 1, dbg);
  HIPCHK(c, hipGetLastError());
  return LCS_OK;
}
int lcs_launch_foe_only(lcs_ctx *c, uint32_t n_cap) {
  int rc = sf_attr(c);
  if (rc) return rc;
  hipLaunchKernelGGL(k_sss_foe, dim3(1, 1), dim3(SF_THREADS), sizeof(SfShared), c->stream, c->peaks, c->npeaks, c->cap32, c->cap64_valid ? c->cap64 : nullptr, n_cap, c->params,
                     0.0, c->d_pss_fd, c->d_sss_fd, 2, (double *)nullptr);
  HIPCHK(c, hipGetLastError());
  return LCS_OK;
}
