/*
 * lcs_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Loop-for-loop C restatement of the LTE-Cell-Scanner searcher hot path.  All
 * "ref:" citations are relative to the reference tree (Evrytania/LTE-Cell-Scanner).
 * See lcs_oracle.h for the pinning status.  Compile with -ffp-contract=off so the
 * arithmetic is the plain IEEE double / float sequence the reference performs.
 */
#include "lcs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define FS_LTE 30720000.0 /* ref: include/constants.h:32 */
#define N_RB_MAXDL 110    /* ref: include/constants.h:33 */
#define PI 3.14159265358979323846

typedef struct { double re, im; } cd;
typedef struct { float re, im; } cf;

static int g_legacy = 0;
static int g_threads = 1;

void orc_set_legacy(int on) { g_legacy = on; }
int orc_get_legacy(void) { return g_legacy; }
void orc_set_threads(int n) { g_threads = n < 1 ? 1 : n; }

/* ---------------------------------------------------------------- helpers */
static inline cd c_(double re, double im) { cd r; r.re = re; r.im = im; return r; }
static inline cd cadd(cd a, cd b) { return c_(a.re + b.re, a.im + b.im); }
static inline cd csub(cd a, cd b) { return c_(a.re - b.re, a.im - b.im); }
/* std::complex operator* for finite values */
static inline cd cmul(cd a, cd b) { return c_(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
static inline cd cconj(cd a) { return c_(a.re, -a.im); }
static inline cd cscale(cd a, double s) { return c_(a.re * s, a.im * s); }
static inline cd cdivr(cd a, double s) { return c_(a.re / s, a.im / s); }
static inline double cabs2(cd a) { return a.re * a.re + a.im * a.im; } /* itpp::sqr(complex) */
static inline double carg_(cd a) { return atan2(a.im, a.re); }
/* std::exp(complex) */
static inline cd cexp_(cd a) { double e = exp(a.re); return c_(e * cos(a.im), e * sin(a.im)); }
/* general complex division (Smith, as libgcc __divdc3 for finite operands) */
static inline cd cdiv(cd x, cd y) {
  double a = x.re, b = x.im, c = y.re, d = y.im, ratio, denom;
  if (fabs(c) < fabs(d)) {
    ratio = c / d; denom = (c * ratio) + d;
    return c_(((a * ratio) + b) / denom, ((b * ratio) - a) / denom);
  }
  ratio = d / c; denom = (d * ratio) + c;
  return c_(((b * ratio) + a) / denom, (b - (a * ratio)) / denom);
}

static inline int round_i(double x) { return (int)rint(x); }  /* itpp::round_i */
static inline int floor_i(double x) { return (int)floor(x); } /* itpp::floor_i */
static inline int imod(int k, int n) {                         /* itpp::mod(int,int) */
  if (n == 0) return k;
  int r = k % n; if (r < 0) r += n; return r;
}
/* ref: include/itpp_ext.h:40-42 */
static inline double matlab_mod(double k, double n) { return (n == 0) ? k : (k - n * floor_i(k / n)); }
/* ref: include/macros.h:49 */
static inline double WRAP(double x, double sm, double lg) { return matlab_mod(x - sm, lg - sm) + sm; }
/* ref: src/itpp_ext.cpp:97-109 -- number of elements of first:incr:last */
static int matlab_range_len(double first, double incr, double last) {
  double s1 = (last - first > 0) - (last - first < 0);
  double s2 = (incr > 0) - (incr < 0);
  if (s1 * s2 >= 0) return floor_i((last - first) / incr) + 1;
  return 0;
}

void orc_cell_init(orc_cell *c) { /* ref: src/common.cpp:36-56 */
  c->fc_requested = NAN; c->fc_programmed = NAN; c->pss_pow = NAN; c->ind = -1; c->freq = NAN;
  c->n_id_2 = -1; c->n_id_1 = -1; c->cp_type = ORC_CP_UNKNOWN; c->frame_start = NAN;
  c->freq_fine = NAN; c->freq_superfine = NAN; c->n_ports = -1; c->n_rb_dl = -1;
  c->phich_duration = 0; c->phich_resource = 0; c->sfn = -1; c->reserved = 0;
}
static inline int cell_n_id_cell(const orc_cell *c) { /* ref: src/common.cpp:29-31 */
  return ((c->n_id_1 >= 0) && (c->n_id_2 >= 0)) ? (c->n_id_2 + 3 * c->n_id_1) : -1;
}
static inline int cell_n_symb_dl(const orc_cell *c) { /* ref: src/common.cpp:32-34 */
  return (c->cp_type == ORC_CP_NORMAL) ? 7 : ((c->cp_type == ORC_CP_EXTENDED) ? 6 : -1);
}

/* ------------------------------------------------------------------- FFT */
/* itpp::fft: X[k] = sum x[n] exp(-j 2 pi k n / N), unscaled. */
static double g_tw_re[64], g_tw_im[64];
static int g_tw_ready = 0;
static void fft128(const cd *in, cd *out) {
  if (!g_tw_ready) {
#pragma omp critical(orc_tw)
    {
      for (int i = 0; i < 64; i++) { g_tw_re[i] = cos(2 * PI * i / 128.0); g_tw_im[i] = -sin(2 * PI * i / 128.0); }
      g_tw_ready = 1;
    }
  }
  for (int i = 0; i < 128; i++) {
    int r = 0;
    for (int b = 0; b < 7; b++) if (i & (1 << b)) r |= 1 << (6 - b);
    out[r] = in[i];
  }
  for (int len = 2; len <= 128; len <<= 1) {
    int half = len >> 1, step = 128 / len;
    for (int s = 0; s < 128; s += len) {
      for (int k = 0; k < half; k++) {
        cd w = c_(g_tw_re[k * step], g_tw_im[k * step]);
        cd u = out[s + k], v = cmul(out[s + k + half], w);
        out[s + k] = cadd(u, v);
        out[s + k + half] = csub(u, v);
      }
    }
  }
}
void orc_fft128(const double *in_re_im, double *out_re_im) { fft128((const cd *)in_re_im, (cd *)out_re_im); }
/* dft(A) = fft(A)/sqrt(length(A))  -- ref: include/dsp.h:34 */
static void dft128(const cd *in, cd *out) {
  fft128(in, out);
  double s = sqrt(128.0);
  for (int i = 0; i < 128; i++) out[i] = cdivr(out[i], s);
}
/* idft(A) = ifft(A)*sqrt(length(A)) -- ref: include/dsp.h:33 ; itpp::ifft scales by 1/N */
static void idft128(const cd *in, cd *out) {
  cd tmp[128], res[128];
  for (int i = 0; i < 128; i++) tmp[i] = cconj(in[i]);
  fft128(tmp, res);
  double s = sqrt(128.0);
  for (int i = 0; i < 128; i++) { cd v = cdivr(cconj(res[i]), 128.0); out[i] = cscale(v, s); }
}

/* ---------------------------------------------------------------- tables */
/* ref: src/lte_lib.cpp:155-161  pss_fd_calc */
static void pss_fd_calc(int t, cd *r /*62*/) {
  static const int zc_map[3] = {25, 29, 34};
  int o = 0;
  for (int n = 0; n < 63; n++) {
    if (n == 31) continue; /* r.del(31) */
    /* exp((complex(0,-1)*pi*zc/63) * (n*(n+1))) */
    cd k = c_(0.0 * PI * zc_map[t] / 63 , -1.0 * PI * zc_map[t] / 63);
    double m = (double)(n * (n + 1));
    r[o++] = cexp_(c_(k.re * m, k.im * m));
  }
}
void orc_pss_fd(int n_id_2, double *out) { pss_fd_calc(n_id_2, (cd *)out); }

/* ref: src/lte_lib.cpp:177-188  PSS_td::PSS_td */
static void pss_td_calc(int t, cd *out /*137*/) {
  cd fd[62], in[128], td[128];
  pss_fd_calc(t, fd);
  memset(in, 0, sizeof(in));
  for (int i = 0; i < 31; i++) in[1 + i] = fd[31 + i];   /* fd(31,61) */
  for (int i = 0; i < 31; i++) in[97 + i] = fd[i];       /* fd(0,30) after 65 zeros */
  idft128(in, td);
  double s = sqrt(128.0 / 62.0);
  for (int i = 0; i < 128; i++) td[i] = cscale(td[i], s);
  for (int i = 0; i < 9; i++) out[i] = td[119 + i];
  for (int i = 0; i < 128; i++) out[9 + i] = td[i];
}
void orc_pss_td(int n_id_2, double *out) { pss_td_calc(n_id_2, (cd *)out); }

/* ref: src/lte_lib.cpp:199-257  sss_fd_calc */
static void sss_fd_calc(int n_id_1, int n_id_2, int slot_num, int32_t *out /*62*/) {
  static const int s_bits[31] = {0,0,0,0,1,0,0,1,0,1,1,0,0,1,1,1,1,1,0,0,0,1,1,0,1,1,1,0,1,0,1};
  static const int c_bits[31] = {0,0,0,0,1,0,1,0,1,1,1,0,1,1,0,0,0,1,1,1,1,1,0,0,1,1,0,1,0,0,1};
  static const int z_bits[31] = {0,0,0,0,1,1,1,0,0,1,1,0,1,1,1,1,1,0,1,0,0,0,1,0,0,1,0,1,0,1,1};
  const int qp = n_id_1 / 30;
  const int q = (n_id_1 + qp * (qp + 1) / 2) / 30;
  const int mp = n_id_1 + q * (q + 1) / 2;
  const int m0 = mp % 31;
  const int m1 = (m0 + mp / 31 + 1) % 31;
  int ssc1[31], ssc2[31];
  for (int n = 0; n < 31; n++) {
    int s0_m0 = 1 - 2 * s_bits[(n + m0) % 31];
    int s1_m1 = 1 - 2 * s_bits[(n + m1) % 31];
    int c0 = 1 - 2 * c_bits[(n + n_id_2) % 31];
    int c1 = 1 - 2 * c_bits[(n + n_id_2 + 3) % 31];
    int z1_m0 = 1 - 2 * z_bits[(n + (m0 % 8)) % 31];
    int z1_m1 = 1 - 2 * z_bits[(n + (m1 % 8)) % 31];
    if (slot_num == 0) { ssc2[n] = s1_m1 * c1 * z1_m0; ssc1[n] = s0_m0 * c0; }
    else               { ssc2[n] = s0_m0 * c1 * z1_m1; ssc1[n] = s1_m1 * c0; }
  }
  for (int n = 0; n < 31; n++) { out[2 * n] = ssc1[n]; out[2 * n + 1] = ssc2[n]; } /* cvectorize of 2x31 */
}
void orc_sss_fd(int n_id_1, int n_id_2, int slot_num, int32_t *out) { sss_fd_calc(n_id_1, n_id_2, slot_num, out); }

/* ref: src/lte_lib.cpp:41-147  lte_pn.  The reference jumps the two LFSRs ahead by
 * 1600 steps with literal GF(2) matrices; clocking them 1600 times is the same map. */
void orc_lte_pn(uint32_t c_init, uint32_t len, uint8_t *out) {
  uint8_t x1[31], x2[31];
  for (int t = 0; t < 31; t++) { x1[t] = 0; x2[t] = (c_init >> t) & 1; }
  x1[0] = 1;
  for (uint32_t t = 0; t < 1600 + len; t++) {
    if (t >= 1600) out[t - 1600] = x1[0] ^ x2[0];
    uint8_t x1n = x1[0] ^ x1[3];
    uint8_t x2n = x2[0] ^ x2[1] ^ x2[2] ^ x2[3];
    memmove(x1, x1 + 1, 30); memmove(x2, x2 + 1, 30);
    x1[30] = x1n; x2[30] = x2n;
  }
}

/* ref: src/lte_lib.cpp:305-324 rs_dl_calc, :327-351 rs_dl_shift_calc, :354-383 RS_DL ctor.
 * rs: [20*n_symb][12]; shift: [20*n_symb][4] (NaN = no RS for that port in that symbol). */
typedef struct { int n_symb_dl; cd rs[20 * 7][12]; double shift[20 * 7][4]; } rs_dl_t;
static void rs_dl_build(int n_id_cell, int cp_type, rs_dl_t *R) {
  const int n_symb_dl = (cp_type == ORC_CP_EXTENDED) ? 6 : 7;
  const int n_rb_dl = 6;
  R->n_symb_dl = n_symb_dl;
  for (int i = 0; i < 20 * 7; i++) {
    for (int k = 0; k < 12; k++) R->rs[i][k] = c_(0, 0);
    for (int p = 0; p < 4; p++) R->shift[i][p] = NAN;
  }
  uint8_t c[4 * N_RB_MAXDL];
  for (int slot_num = 0; slot_num < 20; slot_num++) {
    for (int t = 0; t < 3; t++) {
      int sym_num = (t == 2) ? (n_symb_dl - 3) : t;
      const uint32_t n_cp = (cp_type == ORC_CP_NORMAL);
      const uint32_t c_init = (1u << 10) * (7 * (slot_num + 1) + sym_num + 1) * (2 * n_id_cell + 1) + 2 * n_id_cell + n_cp;
      orc_lte_pn(c_init, 4 * N_RB_MAXDL, c);
      const double isq = 1 / pow(2, 0.5);
      for (int k = 0; k < 2 * n_rb_dl; k++) {
        int m = N_RB_MAXDL - n_rb_dl + k;
        /* (1/sqrt2)*((1-2c(2m)) + J*(1-2c(2m+1))) */
        R->rs[slot_num * n_symb_dl + sym_num][k] = c_(isq * (1 - 2 * c[2 * m]), isq * (1 - 2 * c[2 * m + 1]));
      }
      for (int port = 0; port < 4; port++) {
        double v = NAN;
        if ((port == 0) && (sym_num == 0)) v = 0;
        else if ((port == 0) && (sym_num == n_symb_dl - 3)) v = 3;
        else if ((port == 1) && (sym_num == 0)) v = 3;
        else if ((port == 1) && (sym_num == n_symb_dl - 3)) v = 0;
        else if ((port == 2) && (sym_num == 1)) v = 3 * (slot_num & 1);
        else if ((port == 3) && (sym_num == 1)) v = 3 + 3 * (slot_num & 1);
        int want = ((t == 0) || (t == 2)) ? (port <= 1) : (port >= 2);
        if (want) R->shift[slot_num * n_symb_dl + sym_num][port] = (double)imod((int)(v + n_id_cell), 6);
      }
    }
  }
}
void orc_rs_dl(int n_id_cell, int cp_type, double *rs_re_im, double *shift) {
  rs_dl_t *R = (rs_dl_t *)malloc(sizeof(rs_dl_t));
  rs_dl_build(n_id_cell, cp_type, R);
  int n = 20 * R->n_symb_dl;
  for (int i = 0; i < n; i++) {
    for (int k = 0; k < 12; k++) { rs_re_im[(i * 12 + k) * 2] = R->rs[i][k].re; rs_re_im[(i * 12 + k) * 2 + 1] = R->rs[i][k].im; }
    for (int p = 0; p < 4; p++) shift[i * 4 + p] = R->shift[i][p];
  }
  free(R);
}
static inline const cd *rs_get_rs(const rs_dl_t *R, int slot, int sym) { return R->rs[slot * R->n_symb_dl + sym]; }
static inline double rs_get_shift(const rs_dl_t *R, int slot, int sym, int port) { return R->shift[slot * R->n_symb_dl + sym][port]; }

/* ----------------------------------------------------- chi2cdf_inv (Boost) */
/* ref: include/dsp.h:188-193  chi2cdf_inv(p,k) = 2*gamma_p_inv(k/2,p).  Boost.Math is
 * absent: regularised lower incomplete gamma by series / continued fraction, inverted
 * by bisection-safeguarded Newton to double precision ("parity unpinned": no reference
 * fixture pins this value; checked against scipy in tests). */
static double gamma_p(double a, double x) {
  if (x <= 0) return 0;
  double gln = lgamma(a);
  if (x < a + 1) {
    double ap = a, sum = 1 / a, del = sum;
    for (int n = 0; n < 100000; n++) { ap += 1; del *= x / ap; sum += del; if (fabs(del) < fabs(sum) * 1e-17) break; }
    return sum * exp(-x + a * log(x) - gln);
  }
  double b = x + 1 - a, c = 1 / 1e-300, d = 1 / b, h = d;
  for (int i = 1; i < 100000; i++) {
    double an = -i * (i - a); b += 2; d = an * d + b; if (fabs(d) < 1e-300) d = 1e-300;
    c = b + an / c; if (fabs(c) < 1e-300) c = 1e-300; d = 1 / d; double del = d * c; h *= del;
    if (fabs(del - 1) < 1e-17) break;
  }
  return 1 - exp(-x + a * log(x) - gln) * h;
}
/* upper tail Q(a,x) computed directly (for p close to 1) */
static double gamma_q(double a, double x) {
  if (x <= 0) return 1;
  double gln = lgamma(a);
  if (x < a + 1) return 1 - gamma_p(a, x);
  double b = x + 1 - a, c = 1 / 1e-300, d = 1 / b, h = d;
  for (int i = 1; i < 100000; i++) {
    double an = -i * (i - a); b += 2; d = an * d + b; if (fabs(d) < 1e-300) d = 1e-300;
    c = b + an / c; if (fabs(c) < 1e-300) c = 1e-300; d = 1 / d; double del = d * c; h *= del;
    if (fabs(del - 1) < 1e-17) break;
  }
  return exp(-x + a * log(x) - gln) * h;
}
double orc_chi2cdf_inv(double p, double k) {
  double a = k / 2;
  /* solve Q(a,x) = 1-p with the q formed in double exactly as the reference forms p */
  double q = 1 - p;
  double lo = 0, hi = a + 10 * sqrt(a) + 50;
  while (gamma_q(a, hi) > q) hi *= 2;
  for (int it = 0; it < 400; it++) {
    double mid = 0.5 * (lo + hi);
    if (mid == lo || mid == hi) break;
    if (gamma_q(a, mid) > q) lo = mid; else hi = mid;
  }
  return 2 * 0.5 * (lo + hi);
}

/* ------------------------------------------------------------- fshift etc */
/* ref: include/dsp.h:40-53  fshift */
static void fshift(const cd *seq, int len, double f, double fs, cd *r) {
  double k = PI * f / (fs / 2);
  for (int t = 0; t < len; t++) {
    cd coeff = c_(cos(k * t), sin(k * t));
    r[t] = cmul(seq[t], coeff);
  }
}
/* ref: include/dsp.h:22-29  sigpower */
static double sigpower(const cd *v, int n) {
  double r = 0;
  for (int t = 0; t < n; t++) r += pow(v[t].re, 2) + pow(v[t].im, 2);
  return r / n;
}

/* ------------------------------------------------------------- xcorr_pss */
/* ref: src/searcher.cpp:113-174  xc_correlate.  xc[t][k][foi] as complex<float>. */
static void xc_correlate(const cd *capbuf, uint32_t n_cap, const double *f_search_set, uint32_t n_f,
                         double fc_requested, double fc_programmed, double fs_programmed, cf *xc) {
  const uint32_t n_k = n_cap - 136;
  for (uint32_t foi = 0; foi < n_f; foi++) {
    double f_off = f_search_set[foi];
    double k_factor = (fc_requested - f_off) / fc_programmed;
    for (int t = 0; t < 3; t++) {
      cd pss[137], temp[137];
      pss_td_calc(t, pss);
      /* legacy (MATLAB) mode shifts at the nominal rate: Matlab/xcorr_pss.m:51 */
      fshift(pss, 137, f_off, g_legacy ? fs_programmed : fs_programmed * k_factor, temp);
      for (int m = 0; m < 137; m++) temp[m] = cdivr(cconj(temp[m]), 137);
      long k;
#pragma omp parallel for num_threads(g_threads) schedule(static)
      for (k = 0; k < (long)n_k; k++) {
        double acc_re = 0, acc_im = 0;
        for (int m = 0; m < 137; m++) {
          const cd a = temp[m], b = capbuf[k + m];
          acc_re += a.re * b.re - a.im * b.im;
          acc_im += a.re * b.im + a.im * b.re;
        }
        cf *o = &xc[((size_t)t * n_k + k) * n_f + foi];
        o->re = (float)acc_re; o->im = (float)acc_im;
      }
    }
  }
}

/* ref: src/searcher.cpp:185-221  sp_est */
static void sp_est(const cd *capbuf, uint32_t n_cap, double *sp, double *sp_incoherent, uint16_t *n_comb_sp) {
  *n_comb_sp = (uint16_t)floor_i((n_cap - 136 - 137) / 9600);
  const uint32_t n_sp = *n_comb_sp * 9600;
  sp[0] = 0;
  for (int t = 0; t < 274; t++) sp[0] += pow(capbuf[t].re, 2) + pow(capbuf[t].im, 2);
  sp[0] = sp[0] / 274;
  for (uint32_t t = 1; t < n_sp; t++) {
    sp[t] = sp[t - 1] + (-pow(capbuf[t - 1].re, 2) - pow(capbuf[t - 1].im, 2) + pow(capbuf[t + 274 - 1].re, 2) + pow(capbuf[t + 274 - 1].im, 2)) / 274;
  }
  double tmp[9600];
  for (int i = 0; i < 9600; i++) tmp[i] = sp[i];
  for (int t = 1; t < *n_comb_sp; t++) for (int i = 0; i < 9600; i++) tmp[i] += sp[t * 9600 + i];
  for (int i = 0; i < 9600; i++) tmp[i] = tmp[i] / *n_comb_sp;
  /* tshift(sp_incoherent,137): cyclic right rotate (ref: include/dsp.h:75-87) */
  for (int i = 0; i < 9600; i++) sp_incoherent[(i + 137) % 9600] = tmp[i];
}

/* ref: src/searcher.cpp:263-308  xc_combine */
static void xc_combine(const cf *xc, uint32_t n_k, double fc_requested, double fc_programmed, double fs_programmed,
                       const double *f_search_set, uint32_t n_f, float *single, uint16_t *n_comb_xc) {
  *n_comb_xc = (uint16_t)floor_i((n_k - 100) / 9600);
  for (uint32_t foi = 0; foi < n_f; foi++) {
    const double f_off = f_search_set[foi];
    const double k_factor = (fc_requested - f_off) / fc_programmed;
    for (int t = 0; t < 3; t++) {
      for (int idx = 0; idx < 9600; idx++) single[((size_t)t * 9600 + idx) * n_f + foi] = 0;
      for (int m = 0; m < *n_comb_xc; m++) {
        double actual_start_index = round_i(m * .005 * k_factor * fs_programmed);
        for (int idx = 0; idx < 9600; idx++) {
          const cf v = xc[((size_t)t * n_k + (size_t)(idx + actual_start_index)) * n_f + foi];
          /* sqr(complex<float>) promotes to complex<double>; float += double */
          double s = (double)v.re * (double)v.re + (double)v.im * (double)v.im;
          float *o = &single[((size_t)t * 9600 + idx) * n_f + foi];
          *o = (float)((double)*o + s);
        }
      }
      for (int idx = 0; idx < 9600; idx++) {
        float *o = &single[((size_t)t * 9600 + idx) * n_f + foi];
        *o = *o / (float)(*n_comb_xc);
      }
    }
  }
}

/* ref: src/searcher.cpp:312-347  xc_delay_spread */
static void xc_delay_spread(const float *single, uint32_t n_f, uint32_t ds_comb_arm, float *inc) {
  for (uint32_t foi = 0; foi < n_f; foi++) {
    for (int t = 0; t < 3; t++)
      for (int idx = 0; idx < 9600; idx++) inc[((size_t)t * 9600 + idx) * n_f + foi] = single[((size_t)t * 9600 + idx) * n_f + foi];
    for (uint32_t t = 1; t <= ds_comb_arm; t++) {
      for (int k = 0; k < 3; k++) {
        for (int idx = 0; idx < 9600; idx++) {
          int a = imod(idx - (int)t, 9600), b = imod(idx + (int)t, 9600);
          float s = single[((size_t)k * 9600 + a) * n_f + foi] + single[((size_t)k * 9600 + b) * n_f + foi];
          inc[((size_t)k * 9600 + idx) * n_f + foi] += s;
        }
      }
    }
    for (int t = 0; t < 3; t++)
      for (int idx = 0; idx < 9600; idx++) {
        float *o = &inc[((size_t)t * 9600 + idx) * n_f + foi];
        *o = *o / (float)(2 * ds_comb_arm + 1);
      }
  }
}

/* ref: src/searcher.cpp:353-383  xc_peak_freq */
static void xc_peak_freq(const float *inc, uint32_t n_f, double *pow_, int32_t *frq) {
  for (int t = 0; t < 3; t++) {
    for (int k = 0; k < 9600; k++) {
      const float *row = &inc[((size_t)t * 9600 + k) * n_f];
      double best_pow = row[0];
      uint16_t best_idx = 0;
      for (uint32_t foi = 1; foi < n_f; foi++) {
        if (row[foi] > best_pow) { best_pow = row[foi]; best_idx = (uint16_t)foi; }
      }
      pow_[t * 9600 + k] = best_pow;
      frq[t * 9600 + k] = best_idx;
    }
  }
}

/* ref: src/searcher.cpp:389-419  xcorr_pss */
int orc_xcorr_pss(const double *capbuf_re_im, uint32_t n_cap, const double *f_search_set, uint32_t n_f,
                  uint32_t ds_comb_arm, double fc_requested, double fc_programmed, double fs_programmed,
                  double *pow_, int32_t *frq, float *single, float *incoherent, double *sp_incoherent,
                  float *xc_re_im, double *sp, uint16_t *n_comb_xc, uint16_t *n_comb_sp) {
  const cd *capbuf = (const cd *)capbuf_re_im;
  if (n_cap < 136 + 137 + 9600 || n_f < 1) return -1;
  const uint32_t n_k = n_cap - 136;
  cf *xc = xc_re_im ? (cf *)xc_re_im : (cf *)malloc(sizeof(cf) * 3 * (size_t)n_k * n_f);
  float *inc = incoherent ? incoherent : (float *)malloc(sizeof(float) * 3 * 9600 * (size_t)n_f);
  uint16_t ncsp = (uint16_t)floor_i((n_cap - 136 - 137) / 9600);
  double *spbuf = sp ? sp : (double *)malloc(sizeof(double) * ncsp * 9600);
  if (!xc || !inc || !spbuf) return -2;
  xc_correlate(capbuf, n_cap, f_search_set, n_f, fc_requested, fc_programmed, fs_programmed, xc);
  xc_combine(xc, n_k, fc_requested, fc_programmed, fs_programmed, f_search_set, n_f, single, n_comb_xc);
  xc_delay_spread(single, n_f, ds_comb_arm, inc);
  sp_est(capbuf, n_cap, spbuf, sp_incoherent, n_comb_sp);
  xc_peak_freq(inc, n_f, pow_, frq);
  if (!xc_re_im) free(xc);
  if (!incoherent) free(inc);
  if (!sp) free(spbuf);
  return 0;
}

/* ------------------------------------------------------------ peak_search */
/* ref: src/searcher.cpp:422-510 */
int orc_peak_search(const double *pow_, const int32_t *frq, const double *Z_th1, const double *f_search_set,
                    uint32_t n_f, double fc_requested, double fc_programmed, const float *single,
                    uint32_t ds_comb_arm, orc_cell *cells, int max_cells, int *n_cells) {
  double *w = (double *)malloc(sizeof(double) * 3 * 9600);
  memcpy(w, pow_, sizeof(double) * 3 * 9600);
  int n = 0;
  for (;;) {
    /* max(transpose(working),peak_ind_v): per-PSS first argmax; then first max over the 3 */
    int peak_ind_v[3]; double peak_pow_v[3];
    for (int r = 0; r < 3; r++) {
      double mx = w[r * 9600]; int mi = 0;
      for (int c = 1; c < 9600; c++) if (w[r * 9600 + c] > mx) { mx = w[r * 9600 + c]; mi = c; }
      peak_pow_v[r] = mx; peak_ind_v[r] = mi;
    }
    int peak_n_id_2 = 0; double peak_pow = peak_pow_v[0];
    for (int r = 1; r < 3; r++) if (peak_pow_v[r] > peak_pow) { peak_pow = peak_pow_v[r]; peak_n_id_2 = r; }
    int peak_ind = peak_ind_v[peak_n_id_2];
    if (peak_pow < Z_th1[peak_ind]) break;

    /* refine within +-ds_comb_arm (uint16 loop variable: quirk Q2) */
    double best_pow = -INFINITY;
    int16_t best_ind = -1;
    const int fi = frq[peak_n_id_2 * 9600 + peak_ind];
    for (uint16_t t = (uint16_t)(peak_ind - (int)ds_comb_arm); (int)t <= peak_ind + (int)ds_comb_arm; t++) {
      uint16_t t_wrap = (uint16_t)imod(t, 9600);
      float v = single[((size_t)peak_n_id_2 * 9600 + t_wrap) * n_f + fi];
      if (v > best_pow) { best_pow = v; best_ind = (int16_t)t_wrap; }
    }
    if (n < max_cells) {
      orc_cell c; orc_cell_init(&c);
      c.fc_requested = fc_requested; c.fc_programmed = fc_programmed; c.pss_pow = peak_pow;
      c.ind = best_ind; c.freq = f_search_set[fi]; c.n_id_2 = peak_n_id_2;
      cells[n] = c;
    }
    n++;
    for (int t = -274; t <= 274; t++) w[peak_n_id_2 * 9600 + imod(peak_ind + t, 9600)] = 0;
    /* ref :487-497 "cancel other PSS" loop indexes the already-zeroed row: a no-op (quirk Q1). */
    double thresh = peak_pow * pow(10.0, -12.0 / 10.0);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 9600; c++) if (w[r * 9600 + c] < thresh) w[r * 9600 + c] = 0;
    if (n > 3 * 9600) break; /* safety */
  }
  free(w);
  *n_cells = n;
  return (n > max_cells) ? 1 : 0;
}

/* capbuf.mid(start,128): IT++ does not bounds-check in release builds; the oracle
 * zero-fills samples beyond the buffer instead of reading out of bounds. */
static const cd *mid128(const cd *capbuf, uint32_t n_cap, long start, cd *tmp) {
  if (start >= 0 && (uint64_t)start + 128 <= n_cap) return capbuf + start;
  for (int i = 0; i < 128; i++) { long j = start + i; tmp[i] = (j >= 0 && (uint64_t)j < n_cap) ? capbuf[j] : c_(0, 0); }
  return tmp;
}

/* ----------------------------------------------------------- extract_psss */
/* ref: src/searcher.cpp:516-530 */
static void extract_psss(const cd *td_samps, double foc_freq, double k_factor, double fs_programmed, cd *out /*62*/) {
  cd a[128], b[128], o[128];
  fshift(td_samps, 128, foc_freq, g_legacy ? fs_programmed : fs_programmed * k_factor, a);
  for (int i = 0; i < 126; i++) b[i] = a[i + 2];
  b[126] = a[0]; b[127] = a[1];
  dft128(b, o);
  for (int i = 0; i < 31; i++) out[i] = o[97 + i];
  for (int i = 0; i < 31; i++) out[31 + i] = o[1 + i];
}

/* ref: src/searcher.cpp:533-632  sss_detect_getce_sss */
#define MAX_HF 20
static int sss_detect_getce_sss(const orc_cell *cell, const cd *capbuf, uint32_t n_cap, double fc_requested,
                                double fc_programmed, double fs_programmed, double *h1_np, double *h2_np,
                                cd *h1_nrm, cd *h2_nrm, cd *h1_ext, cd *h2_ext, double *peak_loc_out) {
  double peak_loc = cell->ind;
  const double peak_freq = cell->freq;
  const int n_id_2_est = cell->n_id_2;
  const double k_factor = (fc_requested - peak_freq) / fc_programmed;
  if (peak_loc + 9 < 162) peak_loc += 9600 * k_factor;
  if (peak_loc_out) *peak_loc_out = peak_loc;
  int n_pss = matlab_range_len(peak_loc, k_factor * 9600, (double)n_cap - 125 - 9);
  if (n_pss > MAX_HF) n_pss = MAX_HF;
  if (n_pss < 1) return -1;
  static cd pss_fd_tab[3][62]; static int pss_fd_ready = 0;
  if (!pss_fd_ready) {
#pragma omp critical(orc_pssfd)
    { for (int t = 0; t < 3; t++) pss_fd_calc(t, pss_fd_tab[t]); pss_fd_ready = 1; }
  }
  double pss_np[MAX_HF];
  cd h_raw[MAX_HF][62], h_sm[MAX_HF][62], sss_nrm_raw[MAX_HF][62], sss_ext_raw[MAX_HF][62];
  for (int k = 0; k < n_pss; k++) {
    uint32_t pss_loc = (uint32_t)round_i(peak_loc + k * (k_factor * 9600));
    uint32_t pss_dft_location = pss_loc + 9 - 2;
    cd e[62], w128[128];
    extract_psss(mid128(capbuf, n_cap, (long)pss_dft_location, w128), -peak_freq, k_factor, fs_programmed, e);
    for (int t = 0; t < 62; t++) h_raw[k][t] = cmul(e[t], cconj(pss_fd_tab[n_id_2_est][t]));
    for (int t = 0; t < 62; t++) {
      int lt = (t - 6 > 0) ? t - 6 : 0, rt = (t + 6 < 61) ? t + 6 : 61;
      cd s = c_(0, 0);
      for (int i = lt; i <= rt; i++) s = cadd(s, h_raw[k][i]);
      h_sm[k][t] = cdivr(s, (double)(rt - lt + 1));
    }
    cd d[62];
    for (int t = 0; t < 62; t++) d[t] = csub(h_sm[k][t], h_raw[k][t]);
    pss_np[k] = sigpower(d, 62);
    uint32_t sss_dft_location = pss_dft_location - 128 - 32;
    extract_psss(mid128(capbuf, n_cap, (long)sss_dft_location, w128), -peak_freq, k_factor, fs_programmed, sss_ext_raw[k]);
    sss_dft_location = pss_dft_location - 128 - 9;
    extract_psss(mid128(capbuf, n_cap, (long)sss_dft_location, w128), -peak_freq, k_factor, fs_programmed, sss_nrm_raw[k]);
  }
  for (int t = 0; t < 62; t++) {
    for (int h = 0; h < 2; h++) {
      double s = 0;
      for (int k = h; k < n_pss; k += 2) s += cabs2(h_sm[k][t]) * (1.0 / pss_np[k]);
      double np_est = 1 / (1 + s);
      cd sn = c_(0, 0), se = c_(0, 0);
      for (int k = h; k < n_pss; k += 2) {
        cd w = cmul(cconj(h_sm[k][t]), c_(1.0 / pss_np[k], 0));
        sn = cadd(sn, cmul(w, sss_nrm_raw[k][t]));
        se = cadd(se, cmul(w, sss_ext_raw[k][t]));
      }
      if (h == 0) { h1_np[t] = np_est; h1_nrm[t] = cscale(sn, np_est); h1_ext[t] = cscale(se, np_est); }
      else        { h2_np[t] = np_est; h2_nrm[t] = cscale(sn, np_est); h2_ext[t] = cscale(se, np_est); }
    }
  }
  return 0;
}

/* ref: src/searcher.cpp:636-652  sss_detect_ml_helper */
static double sss_detect_ml_helper(const double *np, const cd *est, const int32_t *try_orig /*124*/) {
  cd acc = c_(0, 0);
  for (int i = 0; i < 124; i++) acc = cadd(acc, cmul(cconj(est[i]), c_((double)try_orig[i], 0)));
  double ang = carg_(acc);
  cd rot = cexp_(cmul(c_(0, 1), c_(-ang, 0)));
  double s1 = 0, s2 = 0;
  cd diff[124];
  for (int i = 0; i < 124; i++) diff[i] = csub(cmul(c_((double)try_orig[i], 0), rot), est[i]);
  for (int i = 0; i < 124; i++) s1 += (diff[i].re * diff[i].re) / np[i];
  for (int i = 0; i < 124; i++) s2 += (diff[i].im * diff[i].im) / np[i];
  return -s1 - s2;
}

/* ref: src/searcher.cpp:696-761  sss_detect (+ :655-693 sss_detect_ml) */
int orc_sss_detect(const orc_cell *cell, const double *capbuf_re_im, uint32_t n_cap, double thresh2_n_sigma,
                   double fc_requested, double fc_programmed, double fs_programmed, orc_cell *cell_out,
                   double *o_h1_np, double *o_h2_np, double *o_h1_nrm, double *o_h2_nrm, double *o_h1_ext,
                   double *o_h2_ext, double *o_ll_nrm, double *o_ll_ext) {
  const cd *capbuf = (const cd *)capbuf_re_im;
  double np12[124]; cd nrm12[124], ext12[124];
  double peak_loc;
  *cell_out = *cell;
  if (sss_detect_getce_sss(cell, capbuf, n_cap, fc_requested, fc_programmed, fs_programmed, np12, np12 + 62,
                           nrm12, nrm12 + 62, ext12, ext12 + 62, &peak_loc) != 0)
    return -1;
  double ll_nrm[168][2], ll_ext[168][2];
  for (int t = 0; t < 168; t++) {
    int32_t h1[62], h2[62], t12[124], t21[124];
    sss_fd_calc(t, cell->n_id_2, 0, h1);
    sss_fd_calc(t, cell->n_id_2, 10, h2);
    for (int i = 0; i < 62; i++) { t12[i] = h1[i]; t12[62 + i] = h2[i]; t21[i] = h2[i]; t21[62 + i] = h1[i]; }
    ll_nrm[t][0] = sss_detect_ml_helper(np12, nrm12, t12);
    ll_nrm[t][1] = sss_detect_ml_helper(np12, nrm12, t21);
    ll_ext[t][0] = sss_detect_ml_helper(np12, ext12, t12);
    ll_ext[t][1] = sss_detect_ml_helper(np12, ext12, t21);
  }
  double mx_n = ll_nrm[0][0], mx_e = ll_ext[0][0];
  for (int t = 0; t < 168; t++) for (int c = 0; c < 2; c++) {
    if (ll_nrm[t][c] > mx_n) mx_n = ll_nrm[t][c];
    if (ll_ext[t][c] > mx_e) mx_e = ll_ext[t][c];
  }
  double (*ll)[2]; int cp_type;
  if (mx_n > mx_e) { ll = ll_nrm; cp_type = ORC_CP_NORMAL; } else { ll = ll_ext; cp_type = ORC_CP_EXTENDED; }

  const double k_factor = (fc_requested - cell->freq) / fc_programmed;
  double frame_start;
  double mx0 = ll[0][0], mx1 = ll[0][1];
  for (int t = 1; t < 168; t++) { if (ll[t][0] > mx0) mx0 = ll[t][0]; if (ll[t][1] > mx1) mx1 = ll[t][1]; }
  int col;
  if (!g_legacy) {
    frame_start = cell->ind + (128 + 9 - 960 - 2) * 16 / FS_LTE * fs_programmed * k_factor;
    if (mx0 > mx1) col = 0;
    else { col = 1; frame_start = frame_start + 9600 * k_factor * 16 / FS_LTE * fs_programmed * k_factor; }
    frame_start = WRAP(frame_start, -0.5, (2 * 9600.0 - 0.5) * 16 / FS_LTE * fs_programmed * k_factor);
  } else { /* Matlab/sss_detect.m:161-168 (0-based) */
    frame_start = peak_loc + (128 + 9 - 960 - 2) * k_factor;
    if (mx0 > mx1) col = 0; else { col = 1; frame_start = frame_start + 9600 * k_factor; }
    frame_start = WRAP(frame_start, -0.5, 2 * 9600.0 - 0.5);
  }
  int n_id_1_est = 0; double lik_final = ll[0][col];
  for (int t = 1; t < 168; t++) if (ll[t][col] > lik_final) { lik_final = ll[t][col]; n_id_1_est = t; }

  /* L = concat(cvectorize(nrm), cvectorize(ext)); itpp::mean / itpp::variance (unbiased) */
  double sum = 0, sq = 0;
  for (int m = 0; m < 2; m++) {
    double (*M)[2] = m ? ll_ext : ll_nrm;
    for (int c = 0; c < 2; c++) for (int t = 0; t < 168; t++) { sum += M[t][c]; sq += M[t][c] * M[t][c]; }
  }
  const int len = 672;
  double lik_mean = sum / len;
  double lik_var = (sq - sum * sum / len) / (len - 1);
  if (lik_final >= lik_mean + pow(lik_var, 0.5) * thresh2_n_sigma) {
    cell_out->n_id_1 = n_id_1_est; cell_out->cp_type = cp_type; cell_out->frame_start = frame_start;
  }
  if (o_h1_np) memcpy(o_h1_np, np12, 62 * sizeof(double));
  if (o_h2_np) memcpy(o_h2_np, np12 + 62, 62 * sizeof(double));
  if (o_h1_nrm) memcpy(o_h1_nrm, nrm12, 62 * sizeof(cd));
  if (o_h2_nrm) memcpy(o_h2_nrm, nrm12 + 62, 62 * sizeof(cd));
  if (o_h1_ext) memcpy(o_h1_ext, ext12, 62 * sizeof(cd));
  if (o_h2_ext) memcpy(o_h2_ext, ext12 + 62, 62 * sizeof(cd));
  if (o_ll_nrm) memcpy(o_ll_nrm, ll_nrm, sizeof(ll_nrm));
  if (o_ll_ext) memcpy(o_ll_ext, ll_ext, sizeof(ll_ext));
  return 0;
}

/* ------------------------------------------------------------ pss_sss_foe */
/* ref: src/searcher.cpp:767-850 */
int orc_pss_sss_foe(const orc_cell *cell_in, const double *capbuf_re_im, uint32_t n_cap, double fc_requested,
                    double fc_programmed, double fs_programmed, orc_cell *cell_out) {
  const cd *capbuf = (const cd *)capbuf_re_im;
  const double k_factor = (fc_requested - cell_in->freq) / fc_programmed;
  uint16_t pss_sss_dist; double first_sss_dft_location;
  *cell_out = *cell_in;
  if (cell_in->cp_type == ORC_CP_NORMAL) {
    pss_sss_dist = (uint16_t)(g_legacy ? round_i((128 + 9) * k_factor) : round_i((128 + 9) * 16 / FS_LTE * fs_programmed * k_factor));
    first_sss_dft_location = cell_in->frame_start + (960 - 128 - 9 - 128) * (g_legacy ? k_factor : 16 / FS_LTE * fs_programmed * k_factor);
  } else if (cell_in->cp_type == ORC_CP_EXTENDED) {
    pss_sss_dist = (uint16_t)round_i((128 + 32) * k_factor); /* quirk Q4 */
    first_sss_dft_location = cell_in->frame_start + (960 - 128 - 32 - 128) * (g_legacy ? k_factor : 16 / FS_LTE * fs_programmed * k_factor);
  } else return -1;
  uint8_t sn;
  first_sss_dft_location = WRAP(first_sss_dft_location, -0.5, 9600 * 2 - 0.5);
  if (first_sss_dft_location - 9600 * k_factor > -0.5) { first_sss_dft_location -= 9600 * k_factor; sn = 10; } else sn = 0;
  const double step = g_legacy ? 9600 * k_factor : 9600 * 16 / FS_LTE * fs_programmed * k_factor;
  int n_sss = matlab_range_len(first_sss_dft_location, step, (double)((int)n_cap - 127 - pss_sss_dist - 100));
  static cd pss_fd_tab[3][62]; static int ready = 0;
  if (!ready) {
#pragma omp critical(orc_pssfd2)
    { for (int t = 0; t < 3; t++) pss_fd_calc(t, pss_fd_tab[t]); ready = 1; }
  }
  sn = (1 - (sn / 10)) * 10;
  cd M = c_(0, 0);
  for (int k = 0; k < n_sss; k++) {
    sn = (1 - (sn / 10)) * 10;
    uint32_t sss_dft_location = (uint32_t)round_i(first_sss_dft_location + k * step);
    uint32_t pss_dft_location = sss_dft_location + pss_sss_dist;
    cd h_raw[62], h_sm[62], sss_raw[62], e[62], w128[128];
    extract_psss(mid128(capbuf, n_cap, (long)pss_dft_location, w128), -cell_in->freq, k_factor, fs_programmed, e);
    for (int t = 0; t < 62; t++) h_raw[t] = cmul(e[t], cconj(pss_fd_tab[cell_in->n_id_2][t]));
    for (int t = 0; t < 62; t++) {
      int lt = (t - 6 > 0) ? t - 6 : 0, rt = (t + 6 < 61) ? t + 6 : 61;
      cd s = c_(0, 0);
      for (int i = lt; i <= rt; i++) s = cadd(s, h_raw[i]);
      h_sm[t] = cdivr(s, (double)(rt - lt + 1));
    }
    cd d[62];
    for (int t = 0; t < 62; t++) d[t] = csub(h_sm[t], h_raw[t]);
    double pss_np = sigpower(d, 62);
    extract_psss(mid128(capbuf, n_cap, (long)sss_dft_location, w128), -cell_in->freq, k_factor, fs_programmed, e);
    /* exp(J*pi*-freq/(FS_LTE/16/2)*-pss_sss_dist) evaluated left to right */
    cd ph = cmul(c_(0, 1), c_(PI, 0));
    ph = cscale(ph, -cell_in->freq);
    ph = cdivr(ph, (FS_LTE / 16 / 2));
    ph = cscale(ph, (double)(-(int)pss_sss_dist));
    ph = cexp_(ph);
    int32_t sfd[62];
    sss_fd_calc(cell_in->n_id_1, cell_in->n_id_2, sn, sfd);
    for (int t = 0; t < 62; t++) sss_raw[t] = cmul(cmul(e[t], ph), c_((double)sfd[t], 0));
    cd acc = c_(0, 0);
    for (int t = 0; t < 62; t++) {
      double a2 = cabs2(h_sm[t]);
      double w = a2 * (1.0 / (2 * a2 * pss_np + pss_np * pss_np));
      acc = cadd(acc, cmul(cmul(cconj(sss_raw[t]), h_raw[t]), c_(w, 0)));
    }
    M = cadd(M, acc);
  }
  if (g_legacy) cell_out->freq_fine = cell_in->freq + carg_(M) / (2 * PI) / (1 / (fs_programmed) * pss_sss_dist);
  else cell_out->freq_fine = cell_in->freq + carg_(M) / (2 * PI) / (1 / (fs_programmed * k_factor) * pss_sss_dist);
  return 0;
}

/* ------------------------------------------------------------ extract_tfg */
/* ref: src/searcher.cpp:857-935 */
int orc_extract_tfg(const orc_cell *cell, const double *capbuf_re_im, uint32_t n_cap, double fc_requested,
                    double fc_programmed, double fs_programmed, double *tfg_re_im, double *tfg_timestamp, int *n_ofdm_out) {
  const cd *capbuf_raw = (const cd *)capbuf_re_im;
  const double frame_start = cell->frame_start;
  const double freq_fine = cell->freq_fine;
  const double k_factor = (fc_requested - cell->freq_fine) / fc_programmed;
  const int n_symb_dl = cell_n_symb_dl(cell);
  double dft_location;
  if (cell->cp_type == ORC_CP_NORMAL) dft_location = frame_start + 10 * 16 / FS_LTE * fs_programmed * k_factor;
  else if (cell->cp_type == ORC_CP_EXTENDED) dft_location = frame_start + 32 * 16 / FS_LTE * fs_programmed * k_factor;
  else return -1;
  if (dft_location - .01 * fs_programmed * k_factor > -0.5) dft_location = dft_location - .01 * fs_programmed * k_factor;
  cd *capbuf = (cd *)malloc(sizeof(cd) * n_cap);
  fshift(capbuf_raw, (int)n_cap, -freq_fine, fs_programmed * k_factor, capbuf);
  const int n_ofdm_sym = 6 * 10 * 2 * n_symb_dl + 2 * n_symb_dl;
  *n_ofdm_out = n_ofdm_sym;
  cd *tfg = (cd *)tfg_re_im;
  int sym_num = 0;
  for (int t = 0; t < n_ofdm_sym; t++) {
    cd o[128];
    int loc = round_i(dft_location);
    if (loc < 0 || (uint32_t)loc + 128 > n_cap) { free(capbuf); return -2; }
    dft128(capbuf + loc, o);
    for (int i = 0; i < 36; i++) tfg[t * 72 + i] = o[92 + i];
    for (int i = 0; i < 36; i++) tfg[t * 72 + 36 + i] = o[1 + i];
    tfg_timestamp[t] = dft_location;
    if (n_symb_dl == 6) dft_location += (128 + 32) * 16 / FS_LTE * fs_programmed * k_factor;
    else {
      if (sym_num == 6) dft_location += (128 + 10) * 16 / FS_LTE * fs_programmed * k_factor;
      else dft_location += (128 + 9) * 16 / FS_LTE * fs_programmed * k_factor;
      sym_num = imod(sym_num + 1, 7);
    }
  }
  for (int t = 0; t < n_ofdm_sym; t++) {
    double ideal_offset = tfg_timestamp[t];
    double actual_offset = round_i(ideal_offset);
    double late = actual_offset - ideal_offset;
    /* exp((-J*2*pi*late/128)*cn) */
    cd k = c_(-0.0, -1.0); k = cscale(k, 2); k = cscale(k, PI); k = cscale(k, late); k = cdivr(k, 128);
    for (int i = 0; i < 72; i++) {
      int cn = (i < 36) ? (i - 36) : (i - 35);
      tfg[t * 72 + i] = cmul(tfg[t * 72 + i], cexp_(c_(k.re * cn, k.im * cn)));
    }
  }
  free(capbuf);
  return 0;
}

/* ------------------------------------------------------------------ tfoec */
/* ref: src/searcher.cpp:952-1069 */
int orc_tfoec(const orc_cell *cell, const double *tfg_re_im, const double *tfg_timestamp, int n_ofdm,
              double fc_requested, double fc_programmed, double *tfg_comp_re_im, double *tfg_comp_timestamp,
              orc_cell *cell_out) {
  const cd *tfg = (const cd *)tfg_re_im;
  cd *tfg_comp = (cd *)tfg_comp_re_im;
  const int n_symb_dl = cell_n_symb_dl(cell);
  if (n_symb_dl < 0) return -1;
  const int n_slot = (int)floor(((double)n_ofdm) / n_symb_dl);
  rs_dl_t *R = (rs_dl_t *)malloc(sizeof(rs_dl_t));
  rs_dl_build(cell_n_id_cell(cell), cell->cp_type, R);
  cd foe = c_(0, 0);
  cd *rs_extracted = (cd *)malloc(sizeof(cd) * n_slot * 12);
  for (int sym_num = 0; sym_num <= n_symb_dl - 3; sym_num += n_symb_dl - 3) {
    for (int t = 0; t < n_slot; t++) {
      int sh = (int)rs_get_shift(R, imod(t, 20), sym_num, 0);
      const cd *rs = rs_get_rs(R, imod(t, 20), sym_num);
      for (int i = 0; i < 12; i++) rs_extracted[t * 12 + i] = cmul(tfg[(t * n_symb_dl + sym_num) * 72 + sh + 6 * i], cconj(rs[i]));
    }
    for (int t = 0; t < 12; t++) {
      cd s = c_(0, 0);
      for (int r = 0; r < n_slot - 1; r++) s = cadd(s, cmul(cconj(rs_extracted[r * 12 + t]), rs_extracted[(r + 1) * 12 + t]));
      foe = cadd(foe, s);
    }
  }
  double residual_f = carg_(foe) / (2 * PI) / 0.0005;
  double k_factor_residual = (fc_requested - residual_f) / fc_programmed;
  for (int t = 0; t < n_ofdm; t++) tfg_comp_timestamp[t] = k_factor_residual * tfg_timestamp[t];
  for (int t = 0; t < n_ofdm; t++) {
    /* exp(J*2*pi*-residual_f*ts_comp/(FS_LTE/16)) */
    cd a = cscale(c_(0, 1), 2); a = cscale(a, PI); a = cscale(a, -residual_f); a = cscale(a, tfg_comp_timestamp[t]); a = cdivr(a, (FS_LTE / 16));
    cd ph = cexp_(a);
    double late = tfg_timestamp[t] - tfg_comp_timestamp[t];
    cd k = c_(-0.0, -1.0); k = cscale(k, 2); k = cscale(k, PI); k = cscale(k, late); k = cdivr(k, 128);
    for (int i = 0; i < 72; i++) {
      int cn = (i < 36) ? (i - 36) : (i - 35);
      cd v = cmul(tfg[t * 72 + i], ph);
      tfg_comp[t * 72 + i] = cmul(v, cexp_(c_(k.re * cn, k.im * cn)));
    }
  }
  cd toe = c_(0, 0);
  for (int t = 0; t < 2 * n_slot - 1; t++) {
    int current_sym_num = (t & 1) ? (n_symb_dl - 3) : 0;
    int current_slot_num = imod((t >> 1), 20);
    int current_offset = (t >> 1) * n_symb_dl + current_sym_num;
    int current_shift = (int)rs_get_shift(R, 0, current_sym_num, 0);
    int next_sym_num = ((t + 1) & 1) ? (n_symb_dl - 3) : 0;
    int next_slot_num = imod(((t + 1) >> 1), 20);
    int next_offset = ((t + 1) >> 1) * n_symb_dl + next_sym_num;
    int next_shift = (int)rs_get_shift(R, 0, next_sym_num, 0);
    int r1_offset, r2_offset, r1_shift, r2_shift, r1_sym, r2_sym, r1_slot, r2_slot;
    if (current_shift < next_shift) {
      r1_offset = current_offset; r1_shift = current_shift; r1_sym = current_sym_num; r1_slot = current_slot_num;
      r2_offset = next_offset; r2_shift = next_shift; r2_sym = next_sym_num; r2_slot = next_slot_num;
    } else {
      r1_offset = next_offset; r1_shift = next_shift; r1_sym = next_sym_num; r1_slot = next_slot_num;
      r2_offset = current_offset; r2_shift = current_shift; r2_sym = current_sym_num; r2_slot = current_slot_num;
    }
    cd r1v[12], r2v[12];
    const cd *rs1 = rs_get_rs(R, r1_slot, r1_sym), *rs2 = rs_get_rs(R, r2_slot, r2_sym);
    for (int i = 0; i < 12; i++) r1v[i] = cmul(tfg_comp[r1_offset * 72 + r1_shift + 6 * i], cconj(rs1[i]));
    for (int i = 0; i < 12; i++) r2v[i] = cmul(tfg_comp[r2_offset * 72 + r2_shift + 6 * i], cconj(rs2[i]));
    cd toe1 = c_(0, 0), toe2 = c_(0, 0);
    for (int i = 0; i < 12; i++) toe1 = cadd(toe1, cmul(cconj(r1v[i]), r2v[i]));
    for (int i = 0; i < 11; i++) toe2 = cadd(toe2, cmul(cconj(r2v[i]), r1v[i + 1]));
    toe = cadd(toe, cadd(toe1, toe2));
  }
  double delay = -carg_(toe) / 3 / (2 * PI / 128);
  cd kk = cscale(c_(0, 1), 2); kk = cscale(kk, PI); kk = cdivr(kk, 128); kk = cscale(kk, delay);
  for (int i = 0; i < 72; i++) {
    int cn = (i < 36) ? (i - 36) : (i - 35);
    cd cv = cexp_(c_(kk.re * cn, kk.im * cn));
    for (int t = 0; t < n_ofdm; t++) tfg_comp[t * 72 + i] = cmul(tfg_comp[t * 72 + i], cv);
  }
  *cell_out = *cell;
  cell_out->freq_superfine = cell_out->freq_fine + residual_f;
  free(rs_extracted); free(R);
  return 0;
}

/* --------------------------------------------------------------- chan_est */
/* ref: include/dsp.h:151-185  interp1 (linear, bisection with round_i midpoint, extrapolating) */
static cd interp1_c(const double *X, const cd *Y, int n, double x) {
  if (n == 1) return Y[0];
  uint32_t try_l = 0, try_r = (uint32_t)n - 1;
  while (try_r - try_l > 1) {
    uint32_t try_mid = (uint32_t)round_i((try_r + try_l) / 2.0);
    if (x >= X[try_mid]) try_l = try_mid; else try_r = try_mid;
  }
  /* Y(l)+(x-X(l))*(Y(r)-Y(l))/(X(r)-X(l)) : double*complex then complex/double */
  cd d = csub(Y[try_r], Y[try_l]);
  cd m = cscale(d, (x - X[try_l]));
  return cadd(Y[try_l], cdivr(m, (X[try_r] - X[try_l])));
}

/* ref: src/searcher.cpp:1200-1213  ce_interp_hex_extend */
static int ce_interp_hex_extend(double *row_x, cd *row_val, int len) {
  if (row_x[0] != 0) {
    cd d = csub(row_val[1], row_val[0]);
    cd v = csub(row_val[0], cdivr(cscale(d, row_x[0]), (row_x[1] - row_x[0])));
    memmove(row_val + 1, row_val, sizeof(cd) * len); memmove(row_x + 1, row_x, sizeof(double) * len);
    row_val[0] = v; row_x[0] = 0; len++;
  }
  if (row_x[len - 1] != 71) {
    cd d = csub(row_val[len - 1], row_val[len - 2]);
    cd v = cadd(row_val[len - 1], cdivr(cscale(d, (71 - row_x[len - 1])), (row_x[len - 1] - row_x[len - 2])));
    row_val[len] = v; row_x[len] = 71; len++;
  }
  return len;
}

/* 3x3 complex solve abc = inv(M)*V via LU with partial pivoting (IT++ inv -> LAPACK zgetrf/zgetri;
 * restated, "parity unpinned" at the ulp level).  ref: src/searcher.cpp:1293-1312 */
static void solve3(cd M[3][3], cd V[3], cd out[3]) {
  cd A[3][4];
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) A[i][j] = M[i][j]; A[i][3] = V[i]; }
  for (int col = 0; col < 3; col++) {
    int piv = col; double best = fabs(A[col][col].re) + fabs(A[col][col].im);
    for (int r = col + 1; r < 3; r++) { double v = fabs(A[r][col].re) + fabs(A[r][col].im); if (v > best) { best = v; piv = r; } }
    if (piv != col) for (int j = 0; j < 4; j++) { cd t = A[col][j]; A[col][j] = A[piv][j]; A[piv][j] = t; }
    for (int r = col + 1; r < 3; r++) {
      cd f = cdiv(A[r][col], A[col][col]);
      for (int j = col; j < 4; j++) A[r][j] = csub(A[r][j], cmul(f, A[col][j]));
    }
  }
  for (int i = 2; i >= 0; i--) {
    cd s = A[i][3];
    for (int j = i + 1; j < 3; j++) s = csub(s, cmul(A[i][j], out[j]));
    out[i] = cdiv(s, A[i][i]);
  }
}

typedef struct { uint8_t x_sc; uint16_t y_symnum; cd val; } tri_vertex;

/* ref: src/searcher.cpp:1223-1362  ce_interp_hex */
static void ce_interp_hex(const cd *ce_filt /*[n_rs][12]*/, const int *shift, int n_ofdm, int n_rs_ofdm,
                          const int *rs_set, cd *ce_tfg /*[n_ofdm][72]*/) {
  for (int t = 0; t <= n_rs_ofdm - 2; t++) {
    double top_row_x[16], bot_row_x[16]; cd top_row_val[16], bot_row_val[16];
    int s_top = (t & 1) ? shift[1] : shift[0], s_bot = (t & 1) ? shift[0] : shift[1];
    int n_top = 0, n_bot = 0;
    for (int x = s_top; x <= 71; x += 6) { top_row_x[n_top] = x; top_row_val[n_top] = ce_filt[t * 12 + n_top]; n_top++; }
    n_top = ce_interp_hex_extend(top_row_x, top_row_val, n_top);
    for (int x = s_bot; x <= 71; x += 6) { bot_row_x[n_bot] = x; bot_row_val[n_bot] = ce_filt[(t + 1) * 12 + n_bot]; n_bot++; }
    n_bot = ce_interp_hex_extend(bot_row_x, bot_row_val, n_bot);
    if (t == 0) for (int x = 0; x <= 71; x++) ce_tfg[rs_set[0] * 72 + x] = interp1_c(top_row_x, top_row_val, n_top, (double)x);

    uint8_t top_row_last_used, bot_row_last_used;
    tri_vertex tri[3];
    if (top_row_x[1] < bot_row_x[1]) {
      tri[0].x_sc = (uint8_t)top_row_x[0]; tri[0].y_symnum = (uint16_t)rs_set[t]; tri[0].val = top_row_val[0];
      tri[1].x_sc = (uint8_t)bot_row_x[0]; tri[1].y_symnum = (uint16_t)rs_set[t + 1]; tri[1].val = bot_row_val[0];
      tri[2].x_sc = (uint8_t)top_row_x[1]; tri[2].y_symnum = (uint16_t)rs_set[t]; tri[2].val = top_row_val[1];
      top_row_last_used = 1; bot_row_last_used = 0;
    } else {
      tri[0].x_sc = (uint8_t)bot_row_x[0]; tri[0].y_symnum = (uint16_t)rs_set[t + 1]; tri[0].val = bot_row_val[0];
      tri[1].x_sc = (uint8_t)top_row_x[0]; tri[1].y_symnum = (uint16_t)rs_set[t]; tri[1].val = top_row_val[0];
      tri[2].x_sc = (uint8_t)bot_row_x[1]; tri[2].y_symnum = (uint16_t)rs_set[t + 1]; tri[2].val = bot_row_val[1];
      top_row_last_used = 0; bot_row_last_used = 1;
    }
    const int spacing = rs_set[t + 1] - rs_set[t];
    double x_offset[16];
    for (int i = 0; i <= spacing; i++) x_offset[i] = 0.0;
    while (1) {
      cd M[3][3], V[3], abc[3];
      for (int i = 0; i < 3; i++) { M[i][0] = c_(tri[i].x_sc, 0); M[i][1] = c_(tri[i].y_symnum, 0); M[i][2] = c_(1, 0); V[i] = tri[i].val; }
      solve3(M, V, abc);
      cd a_p = abc[0], b_p = abc[1], c_p = abc[2];
      double x1 = tri[1].x_sc, x2 = tri[2].x_sc, y1 = tri[1].y_symnum, y2 = tri[2].y_symnum;
      double a_l = (x1 - x2) / (y1 - y2);
      double b_l = (y1 * x2 - y2 * x1) / (y1 - y2);
      for (int r = 1; r <= spacing; r++) {
        while (x_offset[r] <= a_l * (rs_set[t] + r) + b_l) {
          /* a_p*x + b_p*y + c_p  (complex*double) */
          cd v = cadd(cadd(cscale(a_p, x_offset[r]), cscale(b_p, (double)(rs_set[t] + r))), c_p);
          if (x_offset[r] <= 71) ce_tfg[(rs_set[t] + r) * 72 + (int)x_offset[r]] = v;
          x_offset[r]++;
        }
      }
      if ((x_offset[1] == 72) && (x_offset[spacing] == 72)) break;
      if (tri[2].y_symnum == rs_set[t]) {
        tri[0] = tri[1]; tri[1] = tri[2]; bot_row_last_used++;
        if (bot_row_last_used >= n_bot) break; /* defensive: cannot happen for valid shifts */
        tri[2].x_sc = (uint8_t)bot_row_x[bot_row_last_used]; tri[2].y_symnum = (uint16_t)rs_set[t + 1]; tri[2].val = bot_row_val[bot_row_last_used];
      } else {
        tri[0] = tri[1]; tri[1] = tri[2]; top_row_last_used++;
        if (top_row_last_used >= n_top) break;
        tri[2].x_sc = (uint8_t)top_row_x[top_row_last_used]; tri[2].y_symnum = (uint16_t)rs_set[t]; tri[2].val = top_row_val[top_row_last_used];
      }
    }
  }
  for (int t = 0; t < rs_set[0]; t++) memcpy(&ce_tfg[t * 72], &ce_tfg[rs_set[0] * 72], sizeof(cd) * 72);
  for (int t = rs_set[n_rs_ofdm - 1] + 1; t < n_ofdm; t++) memcpy(&ce_tfg[t * 72], &ce_tfg[rs_set[n_rs_ofdm - 1] * 72], sizeof(cd) * 72);
}

static int cmp_int(const void *a, const void *b) { return (*(const int *)a > *(const int *)b) - (*(const int *)a < *(const int *)b); }

/* ref: src/searcher.cpp:1369-1477  chan_est */
typedef struct { cd *raw, *filt; int *rows, *n_rs; } ce_dbg_t;   /* test hook: chan_est's internal estimates */
static int chan_est_x(const orc_cell *cell, const rs_dl_t *R, const cd *tfg, int n_ofdm, int port, cd *ce_tfg, double *np,
                      const ce_dbg_t *dbg) {
  const int n_symb_dl = cell_n_symb_dl(cell);
  int rs_set[512]; int n_rs_ofdm = 0;
  if (port <= 1) {
    for (int x = 0; x <= n_ofdm - 1; x += n_symb_dl) rs_set[n_rs_ofdm++] = x;
    for (int x = n_symb_dl - 3; x <= n_ofdm - 1; x += n_symb_dl) rs_set[n_rs_ofdm++] = x;
    qsort(rs_set, n_rs_ofdm, sizeof(int), cmp_int);
  } else {
    for (int x = 1; x <= n_ofdm - 1; x += n_symb_dl) rs_set[n_rs_ofdm++] = x;
  }
  cd *ce_raw = (cd *)malloc(sizeof(cd) * n_rs_ofdm * 12);
  cd *ce_filt = (cd *)malloc(sizeof(cd) * n_rs_ofdm * 12);
  int slot_num = 0;
  int shift[2] = {-1000, -1000};
  for (int t = 0; t < n_rs_ofdm; t++) {
    int sym_num = imod(rs_set[t], n_symb_dl);
    if (t <= 1) shift[t] = (int)rs_get_shift(R, imod(slot_num, 20), sym_num, port);
    const cd *rs = rs_get_rs(R, slot_num, sym_num);
    int sh = (int)rs_get_shift(R, imod(slot_num, 20), sym_num, port);
    for (int i = 0; i < 12; i++) ce_raw[t * 12 + i] = cmul(tfg[rs_set[t] * 72 + sh + 6 * i], cconj(rs[i]));
    if (((t & 1) == 1) || (port >= 2)) slot_num = imod(slot_num + 1, 20);
  }
  /* 7-point hexagonal mean filter, ref :1431-1467 */
  int current_row_leftmost = shift[0] < shift[1];
  for (int t = 0; t < n_rs_ofdm; t++) {
    for (int k = 0; k < 12; k++) {
      cd total = c_(0, 0); int n_total = 0;
      for (int i = k - 1; i <= k + 1; i++) if (i >= 0 && i <= 11) { total = cadd(total, ce_raw[t * 12 + i]); n_total++; }
      int lo, hi;
      if (shift[0] == shift[1]) { lo = k - 1; hi = k + 1; }
      else if (current_row_leftmost) { lo = k - 1; hi = k; }
      else { lo = k; hi = k + 1; }
      if (t != 0) {
        cd s = c_(0, 0);
        for (int i = lo; i <= hi; i++) if (i >= 0 && i <= 11) { s = cadd(s, ce_raw[(t - 1) * 12 + i]); n_total++; }
        total = cadd(total, s);
      }
      if (t != n_rs_ofdm - 1) {
        cd s = c_(0, 0);
        for (int i = lo; i <= hi; i++) if (i >= 0 && i <= 11) { s = cadd(s, ce_raw[(t + 1) * 12 + i]); n_total++; }
        total = cadd(total, s);
      }
      ce_filt[t * 12 + k] = cdivr(total, (double)n_total);
    }
    current_row_leftmost = !current_row_leftmost;
  }
  /* np = sigpower(cvectorize(ce_filt)-cvectorize(ce_raw)): column-major accumulation order */
  {
    double r = 0;
    for (int k = 0; k < 12; k++) for (int t = 0; t < n_rs_ofdm; t++) {
      cd d = csub(ce_filt[t * 12 + k], ce_raw[t * 12 + k]);
      r += pow(d.re, 2) + pow(d.im, 2);
    }
    *np = r / (n_rs_ofdm * 12);
  }
  if (dbg) {
    memcpy(dbg->raw, ce_raw, sizeof(cd) * n_rs_ofdm * 12);
    memcpy(dbg->filt, ce_filt, sizeof(cd) * n_rs_ofdm * 12);
    memcpy(dbg->rows, rs_set, sizeof(int) * n_rs_ofdm);
    *dbg->n_rs = n_rs_ofdm;
  }
  if (ce_tfg) ce_interp_hex(ce_filt, shift, n_ofdm, n_rs_ofdm, rs_set, ce_tfg);
  free(ce_raw); free(ce_filt);
  return 0;
}
static int chan_est(const orc_cell *cell, const rs_dl_t *R, const cd *tfg, int n_ofdm, int port, cd *ce_tfg, double *np) {
  return chan_est_x(cell, R, tfg, n_ofdm, port, ce_tfg, np, NULL);
}
/* chan_est's raw (ref src/searcher.cpp:1404-1419) and hexagonally filtered (:1431-1467) reference-signal estimates,
 * [n_rs][12] each, and the grid rows they were taken from (<= 244 rows) -- exported so that the tests can pin the
 * tracker restatement below to them. */
int orc_chan_est_dbg(const orc_cell *cell, const double *tfg_re_im, int n_ofdm, int port, double *raw_re_im, double *filt_re_im,
                     int *rows, int *n_rs) {
  rs_dl_t *R = (rs_dl_t *)malloc(sizeof(rs_dl_t));
  rs_dl_build(cell_n_id_cell(cell), cell->cp_type, R);
  ce_dbg_t d = {(cd *)raw_re_im, (cd *)filt_re_im, rows, n_rs};
  double np;
  const int rc = chan_est_x(cell, R, (const cd *)tfg_re_im, n_ofdm, port, NULL, &np, &d);
  free(R);
  return rc;
}
int orc_chan_est(const orc_cell *cell, const double *tfg_re_im, int n_ofdm, int port, double *ce_re_im, double *np) {
  rs_dl_t *R = (rs_dl_t *)malloc(sizeof(rs_dl_t));
  rs_dl_build(cell_n_id_cell(cell), cell->cp_type, R);
  int rc = chan_est(cell, R, (const cd *)tfg_re_im, n_ofdm, port, (cd *)ce_re_im, np);
  free(R);
  return rc;
}

/* ------------------------------------------------------- PBCH decode chain */
/* ref: src/lte_lib.cpp:409-463 lte_conv_ratematch on the probe matrix used by
 * lte_conv_deratematch (:469-518): returns, for each of n_e output positions, the
 * (stream r, column c) it carries. */
static void ratematch_probe(int n_c_cols /*40*/, int n_e, uint8_t *r_of, uint16_t *c_of) {
  static const int perm_pattern[32] = {1,17,9,25,5,21,13,29,3,19,11,27,7,23,15,31,0,16,8,24,4,20,12,28,2,18,10,26,6,22,14,30};
  const int n_c = 32;
  const int n_r = (int)ceil((double)n_c_cols / n_c);
  const int tot = n_r * n_c;
  /* v[t][*]: stream t after sub-block interleaving; -1 marks <NULL> */
  int *v = (int *)malloc(sizeof(int) * 3 * tot);
  for (int t = 0; t < 3; t++) {
    /* temp_row = [NaN x (tot-cols), d(t,:)]; y = transpose(reshape(temp_row,n_c,n_r)) -> y(r,c)=temp_row[r*n_c+c] */
    /* y_perm.col(k) = y.col(perm(k)); v(t,:) = cvectorize(y_perm) -> column-major */
    for (int k = 0; k < n_c; k++) for (int r = 0; r < n_r; r++) {
      int src = r * n_c + perm_pattern[k];
      int col = src - (tot - n_c_cols);
      v[t * tot + k * n_r + r] = (col >= 0) ? col : -1;
    }
  }
  /* w = cvectorize(transpose(v)) = [v(0,:), v(1,:), v(2,:)] */
  int k = 0, j = 0;
  while (k < n_e) {
    int t = j / tot, i = j % tot;
    if (v[t * tot + i] >= 0) { r_of[k] = (uint8_t)t; c_of[k] = (uint16_t)v[t * tot + i]; k++; }
    j = imod(j + 1, 3 * tot);
  }
  free(v);
}

/* ref: src/lte_lib.cpp:469-518  lte_conv_deratematch: d_x[3][n_c] */
static void lte_conv_deratematch(const double *e_est, int n_e, int n_c, double *d_x) {
  uint8_t *r_of = (uint8_t *)malloc(n_e); uint16_t *c_of = (uint16_t *)malloc(sizeof(uint16_t) * n_e);
  int *cnt = (int *)calloc(3 * n_c, sizeof(int));
  ratematch_probe(n_c, n_e, r_of, c_of);
  for (int i = 0; i < 3 * n_c; i++) d_x[i] = 0.0;
  for (int t = 0; t < n_e; t++) { d_x[r_of[t] * n_c + c_of[t]] += e_est[t]; cnt[r_of[t] * n_c + c_of[t]]++; }
  for (int i = 0; i < 3 * n_c; i++) if (cnt[i] > 1) d_x[i] = d_x[i] / cnt[i];
  free(r_of); free(c_of); free(cnt);
}

/* ref: src/lte_lib.cpp:538-551 lte_conv_decode -> itpp Convolutional_Code::decode_tailbite
 * (K=7, generators 0133,0171,0165; exhaustive tail-biting Viterbi: one trellis pass per
 * start state with the end state forced equal, keep the minimum-metric path).  Branch
 * metric: sum_j (out_j ? +r_j : -r_j), minimised, formed as IT++'s calc_metric
 * forms it: the metric of each of the 2^n output words of a step is accumulated on its own,
 * from the last generator's observation to the first, and then added to the path metric
 * (IT++ is not in the reference tree: restated from its published algorithm; pinned at the
 * level of the decoded bits by the reference captures' CRC-passing MIBs). */
static void conv_decode_tailbite(const double *d_est /*[3][n]*/, int n, uint8_t *c_est) {
  static const int G[3] = {0133, 0171, 0165};
  uint8_t outbits[128][3];
  for (int reg = 0; reg < 128; reg++) for (int j = 0; j < 3; j++) outbits[reg][j] = (uint8_t)__builtin_parity(reg & G[j]);
  double best_total = INFINITY;
  uint8_t *best = (uint8_t *)malloc(n), *cur = (uint8_t *)malloc(n);
  uint8_t *surv = (uint8_t *)malloc((size_t)n * 64);
  for (int ss = 0; ss < 64; ss++) {
    double pm[64], npm[64];
    for (int s = 0; s < 64; s++) pm[s] = INFINITY;
    pm[ss] = 0;
    for (int t = 0; t < n; t++) {
      double delta[8];        /* output word w = out_0 + 2 out_1 + 4 out_2 */
      for (int w = 0; w < 8; w++) {
        double dm = 0;
        for (int j = 2; j >= 0; j--) { double r = d_est[j * n + t]; if ((w >> j) & 1) dm += r; else dm -= r; }
        delta[w] = dm;
      }
      for (int s = 0; s < 64; s++) npm[s] = INFINITY;
      for (int s = 0; s < 64; s++) {
        if (pm[s] == INFINITY) continue;
        for (int b = 0; b < 2; b++) {
          int reg = (b << 6) | s;
          int ns = reg >> 1;
          double m = pm[s] + delta[outbits[reg][0] | (outbits[reg][1] << 1) | (outbits[reg][2] << 2)];
          if (m < npm[ns]) { npm[ns] = m; surv[(size_t)t * 64 + ns] = (uint8_t)s; }
        }
      }
      memcpy(pm, npm, sizeof(pm));
    }
    if (pm[ss] < best_total) {
      int s = ss;
      for (int t = n - 1; t >= 0; t--) { cur[t] = (uint8_t)((s >> 5) & 1); s = surv[(size_t)t * 64 + s]; }
      best_total = pm[ss];
      memcpy(best, cur, n);
    }
  }
  memcpy(c_est, best, n);
  free(best); free(cur); free(surv);
}

/* ref: src/lte_lib.cpp:637-663 lte_calc_crc(CRC16): poly 1 0001 0000 0010 0001, zero init */
static void crc16_bits(const uint8_t *a, int n, uint8_t *p /*16*/) {
  static const uint8_t poly[17] = {1,0,0,0,1,0,0,0,0,0,0,1,0,0,0,0,1};
  uint8_t buf[64 + 16];
  memset(buf, 0, sizeof(buf));
  memcpy(buf, a, n);
  for (int i = 0; i < n; i++) if (buf[i]) for (int j = 0; j < 17; j++) buf[i + j] ^= poly[j];
  memcpy(p, buf + n, 16);
}

/* ref: src/lte_lib.cpp:612-634 lte_demodulate(QAM) -> itpp Modulator::demodulate_soft_bits
 * (LOGMAP): llr_i = trunc_log(sum_{s:b_i=0} e^{-|rx-ch*s|^2/N0}) - trunc_log(sum_{s:b_i=1} ...),
 * rx = sym/sqrt(np), ch = 1/sqrt(np), N0 = 1; QPSK map (ref :560-567): 00->(1,1) 01->(1,-1)
 * 10->(-1,1) 11->(-1,-1), all /sqrt(2). */
static double trunc_log(double x) {
  if (x == INFINITY) return log(DBL_MAX);
  if (x <= 0) return log(DBL_MIN);
  return log(x);
}
static void lte_demodulate_qpsk(const cd *syms, const double *np, int n, double *llr) {
  const double a = 1 / sqrt(2.0);
  const cd S[4] = {{a, a}, {a, -a}, {-a, a}, {-a, -a}};
  for (int l = 0; l < n; l++) {
    cd gain = cdiv(c_(1.0, 0), c_(sqrt(np[l]), 0));
    cd rx = cmul(syms[l], gain);
    double metric[4];
    for (int j = 0; j < 4; j++) metric[j] = exp(-cabs2(csub(rx, cmul(gain, S[j]))) / 1);
    llr[2 * l] = trunc_log(metric[0] + metric[1]) - trunc_log(metric[2] + metric[3]);
    llr[2 * l + 1] = trunc_log(metric[0] + metric[2]) - trunc_log(metric[1] + metric[3]);
  }
}

void orc_conv_decode_tailbite(const double *d_est, int n, uint8_t *c_est) { conv_decode_tailbite(d_est, n, c_est); }
/* ref: src/searcher.cpp:1617-1636: CRC-16 of the 24 payload bits, antenna-port mask, compare with bits 24..39 */
int orc_pbch_crc_ok(const uint8_t *c_est, int n_ports) {
  uint8_t crc_est[16];
  crc16_bits(c_est, 24, crc_est);
  if (n_ports == 2) for (int t = 0; t < 16; t++) crc_est[t] = 1 - crc_est[t];
  else if (n_ports == 4) for (int t = 1; t < 16; t += 2) crc_est[t] = 1 - crc_est[t];
  return memcmp(crc_est, c_est + 24, 16) == 0;
}

/* ref: src/searcher.cpp:1482-1522 pbch_extract, :1526-1692 decode_mib */
int orc_decode_mib(const orc_cell *cell, const double *tfg_re_im, int n_ofdm, orc_cell *cell_out) {
  const cd *tfg = (const cd *)tfg_re_im;
  const int n_symb_dl = cell_n_symb_dl(cell);
  *cell_out = *cell;
  if (n_symb_dl < 0) return -1;
  rs_dl_t *R = (rs_dl_t *)malloc(sizeof(rs_dl_t));
  rs_dl_build(cell_n_id_cell(cell), cell->cp_type, R);
  cd *ce[4]; double np_v[4];
  for (int p = 0; p < 4; p++) { ce[p] = (cd *)malloc(sizeof(cd) * n_ofdm * 72); chan_est(cell, R, tfg, n_ofdm, p, ce[p], &np_v[p]); }
  const int m_bit = (cell->cp_type == ORC_CP_NORMAL) ? 1920 : 1728;
  const int n_sym = m_bit / 2;
  const int v_shift_m3 = imod(cell_n_id_cell(cell), 3);
  cd *pbch_sym = (cd *)malloc(sizeof(cd) * n_sym), *pbch_ce = (cd *)malloc(sizeof(cd) * 4 * n_sym);
  cd *syms = (cd *)malloc(sizeof(cd) * n_sym); double *np = (double *)malloc(sizeof(double) * n_sym);
  double *e_est = (double *)malloc(sizeof(double) * m_bit); uint8_t *scr = (uint8_t *)malloc(m_bit);
  int found = 0;
  for (int frame_timing_guess = 0; frame_timing_guess <= 3 && !found; frame_timing_guess++) {
    const int start = frame_timing_guess * 10 * 2 * n_symb_dl;
    int idx = 0;
    for (int fr = 0; fr <= 3; fr++) for (int sym = 0; sym <= 3; sym++) for (int sc = 0; sc <= 71; sc++) {
      if ((imod(sc, 3) == v_shift_m3) && ((sym == 0) || (sym == 1) || ((sym == 3) && (n_symb_dl == 6)))) continue;
      int sym_num = start + fr * 10 * 2 * n_symb_dl + n_symb_dl + sym;
      pbch_sym[idx] = tfg[sym_num * 72 + sc];
      for (int p = 0; p < 4; p++) pbch_ce[p * n_sym + idx] = ce[p][sym_num * 72 + sc];
      idx++;
    }
    for (int n_ports_pre = 1; n_ports_pre <= 3 && !found; n_ports_pre++) {
      const int n_ports = (n_ports_pre == 3) ? 4 : n_ports_pre;
      if (n_ports == 1) {
        for (int t = 0; t < n_sym; t++) {
          cd h = pbch_ce[t];
          cd gain = cconj(cdiv(h, c_(cabs2(h), 0)));
          syms[t] = cmul(pbch_sym[t], gain);
          np[t] = np_v[0] * cabs2(gain);
        }
      } else {
        for (int t = 0; t < n_sym; t += 2) {
          cd h1, h2; double np_temp;
          if (n_ports == 2) {
            h1 = cdivr(cadd(pbch_ce[0 * n_sym + t], pbch_ce[0 * n_sym + t + 1]), 2);
            h2 = cdivr(cadd(pbch_ce[1 * n_sym + t], pbch_ce[1 * n_sym + t + 1]), 2);
            np_temp = (np_v[0] + np_v[1]) / 2; /* mean(np_v(0,1)) */
          } else if (imod(t, 4) == 0) {
            h1 = cdivr(cadd(pbch_ce[0 * n_sym + t], pbch_ce[0 * n_sym + t + 1]), 2);
            h2 = cdivr(cadd(pbch_ce[2 * n_sym + t], pbch_ce[2 * n_sym + t + 1]), 2);
            np_temp = (np_v[0] + np_v[2]) / 2;
          } else {
            h1 = cdivr(cadd(pbch_ce[1 * n_sym + t], pbch_ce[1 * n_sym + t + 1]), 2);
            h2 = cdivr(cadd(pbch_ce[3 * n_sym + t], pbch_ce[3 * n_sym + t + 1]), 2);
            np_temp = (np_v[1] + np_v[3]) / 2;
          }
          cd x1 = pbch_sym[t], x2 = pbch_sym[t + 1];
          double scale = pow(h1.re, 2) + pow(h1.im, 2) + pow(h2.re, 2) + pow(h2.im, 2);
          syms[t] = cdivr(cadd(cmul(cconj(h1), x1), cmul(h2, cconj(x2))), scale);
          cd nh2c = c_(-h2.re, h2.im); /* -conj(h2) */
          syms[t + 1] = cconj(cdivr(cadd(cmul(nh2c, x1), cmul(h1, cconj(x2))), scale));
          np[t] = (pow(hypot(h1.re, h1.im) / scale, 2) + pow(hypot(h2.re, h2.im) / scale, 2)) * np_temp;
          np[t + 1] = np[t];
        }
        double s2 = pow(2, 0.5);
        for (int t = 0; t < n_sym; t++) syms[t] = cscale(syms[t], s2);
      }
      lte_demodulate_qpsk(syms, np, n_sym, e_est);
      orc_lte_pn((uint32_t)cell_n_id_cell(cell), (uint32_t)m_bit, scr);
      for (int t = 0; t < m_bit; t++) if (scr[t]) e_est[t] = -e_est[t];
      double d_est[3 * 40];
      lte_conv_deratematch(e_est, m_bit, 40, d_est);
      uint8_t c_est[40], crc_est[16];
      conv_decode_tailbite(d_est, 40, c_est);
      crc16_bits(c_est, 24, crc_est);
      if (n_ports == 2) for (int t = 0; t < 16; t++) crc_est[t] = 1 - crc_est[t];
      else if (n_ports == 4) for (int t = 1; t < 16; t += 2) crc_est[t] = 1 - crc_est[t];
      if (memcmp(crc_est, c_est + 24, 16) == 0) {
        cell_out->n_ports = n_ports;
        const int bw_packed = c_est[0] * 4 + c_est[1] * 2 + c_est[2];
        static const int bw[6] = {6, 15, 25, 50, 75, 100};
        if (bw_packed < 6) cell_out->n_rb_dl = bw[bw_packed];
        cell_out->phich_duration = c_est[3] ? 2 : 1;
        cell_out->phich_resource = 1 + c_est[4] * 2 + c_est[5];
        int8_t sfn_temp = (int8_t)(128 * c_est[6] + 64 * c_est[7] + 32 * c_est[8] + 16 * c_est[9] + 8 * c_est[10] + 4 * c_est[11] + 2 * c_est[12] + c_est[13]); /* quirk Q10 */
        cell_out->sfn = imod(sfn_temp * 4 - frame_timing_guess, 1024);
        found = 1;
      }
    }
  }
  for (int p = 0; p < 4; p++) free(ce[p]);
  free(pbch_sym); free(pbch_ce); free(syms); free(np); free(e_est); free(scr); free(R);
  return 0;
}

/* ----------------------------------------------------- whole-capbuf chain */
/* ref: src/CellSearch.cpp:484-558 */
int orc_search_capbuf(const double *capbuf_re_im, uint32_t n_cap, const double *f_search_set, uint32_t n_f,
                      double fc_requested, double fc_programmed, double fs_programmed, orc_cell *cells,
                      int max_cells, int *n_cells, orc_cell *peaks, int max_peaks, int *n_peaks) {
  const uint32_t DS_COMB_ARM = 2;
  double *pow_ = (double *)malloc(sizeof(double) * 3 * 9600);
  int32_t *frq = (int32_t *)malloc(sizeof(int32_t) * 3 * 9600);
  float *single = (float *)malloc(sizeof(float) * 3 * 9600 * (size_t)n_f);
  double sp_incoherent[9600], Z_th1[9600];
  uint16_t n_comb_xc, n_comb_sp;
  int rc = orc_xcorr_pss(capbuf_re_im, n_cap, f_search_set, n_f, DS_COMB_ARM, fc_requested, fc_programmed, fs_programmed,
                         pow_, frq, single, NULL, sp_incoherent, NULL, NULL, &n_comb_xc, &n_comb_sp);
  if (rc) { free(pow_); free(frq); free(single); return rc; }
  const int thresh1_n_nines = 12;
  double R_th1 = orc_chi2cdf_inv(1 - pow(10.0, -thresh1_n_nines), 2 * n_comb_xc * (2 * DS_COMB_ARM + 1));
  double rx_cutoff = (6 * 12 * 15e3 / 2 + 4 * 15e3) / (FS_LTE / 16 / 2);
  for (int i = 0; i < 9600; i++) Z_th1[i] = R_th1 * sp_incoherent[i] / rx_cutoff / 137 / 2 / n_comb_xc / (2 * DS_COMB_ARM + 1);
  enum { MAXP = 256 };
  orc_cell *pk = (orc_cell *)malloc(sizeof(orc_cell) * MAXP);
  int npk = 0;
  orc_peak_search(pow_, frq, Z_th1, f_search_set, n_f, fc_requested, fc_programmed, single, DS_COMB_ARM, pk, MAXP, &npk);
  if (npk > MAXP) npk = MAXP;
  if (n_peaks) { *n_peaks = npk; for (int i = 0; i < npk && i < max_peaks; i++) peaks[i] = pk[i]; }
  int n_out = 0;
  double *tfg = (double *)malloc(sizeof(double) * 2 * 854 * 72), *tfgc = (double *)malloc(sizeof(double) * 2 * 854 * 72);
  double ts[854], tsc[854];
  for (int i = 0; i < npk; i++) {
    orc_cell c = pk[i], c2;
    orc_sss_detect(&c, capbuf_re_im, n_cap, 3, fc_requested, fc_programmed, fs_programmed, &c2, 0, 0, 0, 0, 0, 0, 0, 0);
    if (c2.n_id_1 == -1) continue;
    orc_pss_sss_foe(&c2, capbuf_re_im, n_cap, fc_requested, fc_programmed, fs_programmed, &c);
    int n_ofdm = 0;
    if (orc_extract_tfg(&c, capbuf_re_im, n_cap, fc_requested, fc_programmed, fs_programmed, tfg, ts, &n_ofdm)) continue;
    orc_tfoec(&c, tfg, ts, n_ofdm, fc_requested, fc_programmed, tfgc, tsc, &c2);
    orc_decode_mib(&c2, tfgc, n_ofdm, &c);
    if (c.n_rb_dl == -1) continue;
    if (n_out < max_cells) cells[n_out] = c;
    n_out++;
  }
  *n_cells = n_out;
  free(pow_); free(frq); free(single); free(pk); free(tfg); free(tfgc);
  return 0;
}

/* ==================================================================================================
 * LTE-Tracker per-symbol pipeline (reference src/tracker_thread.cpp), restated for a BLOCK of
 * n_sym consecutive OFDM symbols of one tracked cell: symbol i of the block is (slot, sym) = (slot0, sym0) advanced
 * i times by slot_sym_inc (include/LTE-Tracker.h:255-263), with the 128 time-domain samples and the capture metadata
 * (frequency_offset, frame_timing, late) the producer thread queues with every symbol (src/producer_thread.cpp:
 * 196-246).  The reference walks the symbols one at a time through small fifos; the functions below produce the same
 * per-symbol results as arrays.  The slow feedback recurrences those results feed (global frequency offset,
 * src/tracker_thread.cpp:235-242; frame timing, :283-287; the mib_decode_failures state machine, :706-745) are
 * scalar and stay with the caller.
 */
static void slot_sym_inc(int n_symb_dl, int *slot, int *sym) { /* ref: include/LTE-Tracker.h:255-263 */
  *sym = imod(*sym + 1, n_symb_dl);
  if (*sym == 0) *slot = imod(*slot + 1, 20);
}

/* ref: src/tracker_thread.cpp:91-174 get_fd.  td: [n_sym][128] complex; out syms: [n_sym][72]; *bpo is the running
 * bulk_phase_offset (in: value before the block, out: after it); bpo_trace (nullable): its value at every symbol. */
int orc_trk_get_fd(const orc_cell *cell, const double *td_re_im, int n_sym, int slot0, int sym0, const double *freq_off,
                   const double *late, double fc_requested, double fc_programmed, double fs_programmed, double *bpo,
                   double *syms_re_im, double *bpo_trace) {
  const int n_symb_dl = cell_n_symb_dl(cell);
  if (n_symb_dl < 0) return -1;
  const cd *td = (const cd *)td_re_im;
  cd *out = (cd *)syms_re_im;
  int slot = slot0, sym = sym0;
  double bulk_phase_offset = *bpo;
  for (int i = 0; i < n_sym; i++) {
    const double frequency_offset = freq_off[i];
    const double k_factor = (fc_requested - frequency_offset) / fc_programmed;
    cd data[128], dft_in[128], dft_out[128], syms[72];
    fshift(td + (size_t)i * 128, 128, -frequency_offset, fs_programmed * k_factor, data);      /* :126 */
    for (int t = 0; t < 126; t++) dft_in[t] = data[t + 2];                                      /* :129-134 */
    dft_in[126] = data[0];
    dft_in[127] = data[1];
    dft128(dft_in, dft_out);                                                                    /* :135 */
    for (int t = 0; t < 36; t++) { syms[t + 36] = dft_out[t + 1]; syms[t] = dft_out[92 + t]; }  /* :138-141 */
    int n_samp_elapsed;
    if (cell->cp_type == ORC_CP_EXTENDED) n_samp_elapsed = 128 + 32;                            /* :146-150 */
    else n_samp_elapsed = (sym == 0) ? 128 + 10 : 128 + 9;
    const double k = 2 * PI * late[i] / 128;
    bulk_phase_offset = WRAP(bulk_phase_offset + 2 * PI * n_samp_elapsed * (1 / (FS_LTE / 16)) * -frequency_offset, -PI, PI);
    const cd bpo_coeff = c_(cos(bulk_phase_offset), sin(bulk_phase_offset));
    for (int t = 1; t <= 36; t++) {                                                             /* :158-165 */
      const double phase = -k * t;
      cd coeff = c_(cos(phase), sin(phase));
      syms[35 + t] = cmul(syms[35 + t], cmul(bpo_coeff, coeff));
      coeff.im = -coeff.im;
      syms[36 - t] = cmul(syms[36 - t], cmul(bpo_coeff, coeff));
    }
    memcpy(out + (size_t)i * 72, syms, sizeof(syms));
    if (bpo_trace) bpo_trace[i] = bulk_phase_offset;
    slot_sym_inc(n_symb_dl, &slot, &sym);
  }
  *bpo = bulk_phase_offset;
  return 0;
}

typedef struct { double shift; int slot, sym, idx; double frequency_offset, frame_timing; cd ce[12]; } trk_raw_t;
typedef struct { double shift; int slot, sym, idx; double tp, sp, sp_raw, np; cd ce_filt[12]; } trk_filt_t;

/* ref: src/tracker_thread.cpp:176-201 filter_ce (matlab_range + del_oob = the index ranges clipped to 0..11) */
static void trk_filter_ce(const trk_raw_t *prev, const trk_raw_t *curr, const trk_raw_t *next, cd *ce_filt) {
  for (int t = 0; t < 12; t++) {
    cd total = c_(0, 0);
    int n_total = 0;
    for (int k = t - 1; k <= t + 1; k++) if (k >= 0 && k <= 11) { total = cadd(total, curr->ce[k]); n_total++; }
    const int lo = (prev->shift < curr->shift) ? t : t - 1, hi = lo + 1;
    cd sp_ = c_(0, 0), sn_ = c_(0, 0);
    int n_ind = 0;
    for (int k = lo; k <= hi; k++) if (k >= 0 && k <= 11) { sp_ = cadd(sp_, prev->ce[k]); sn_ = cadd(sn_, next->ce[k]); n_ind++; }
    total = cadd(total, sp_);
    total = cadd(total, sn_);
    n_total += 2 * n_ind;
    ce_filt[t] = cdivr(total, (double)n_total);
  }
}

/* ref: src/tracker_thread.cpp:383-400 interp72 */
static void trk_interp72(const trk_filt_t *rs, cd *interp) {
  int l_x = (int)rs->shift;
  cd l_y = rs->ce_filt[0];
  int r_x = (int)rs->shift + 6;
  cd r_y = rs->ce_filt[1];
  int ptr = 1;
  for (int t = 0; t < 72; t++) {
    if ((t > r_x) && (ptr < 11)) { l_x = r_x; l_y = r_y; r_x += 6; ptr++; r_y = rs->ce_filt[ptr]; }
    interp[t] = cadd(cscale(cdivr(csub(r_y, l_y), (double)(r_x - l_x)), (double)(t - l_x)), l_y);
  }
}

/* RS extraction (ref :868-890), filtering + power measurements (:906-931), the FOE and TOE measurements (do_foe
 * :203-243, do_toe_v2 :245-288, up to the value each one feeds into its recurrence) and the 2-D interpolation
 * (interp2d :402-477) for every port of a block.
 *   meas    [4][max_rs][9]: symbol index, np, tp, sp_raw, sp, frequency_offset + residual_f, residual_f_np,
 *                           rs_curr.frame_timing + delay, delay_np -- one row per filtered RS symbol
 *   n_meas  [4]            rows filled per port
 *   ce      [4][n_sym][72] complex, ce_pw [4][n_sym][4] (tp, sp, sp_raw, np), valid for symbols < ce_upto[port]
 */
typedef struct { int port; cd *raw, *filt; int *raw_idx, *filt_idx, *n_raw, *n_filt; } trk_dbg_t;   /* test hook */
static int trk_chan_est_x(const orc_cell *cell, const double *syms_re_im, int n_sym, int slot0, int sym0, const double *freq_off,
                     const double *frame_timing, double fc_requested, double fc_programmed, double fs_programmed,
                     double *meas, int max_rs, int *n_meas, double *ce_re_im, double *ce_pw, int *ce_upto, const trk_dbg_t *dbg) {
  const int n_symb_dl = cell_n_symb_dl(cell);
  if (n_symb_dl < 0 || cell->n_ports < 1 || cell->n_ports > 4) return -1;
  const cd *syms = (const cd *)syms_re_im;
  rs_dl_t *R = (rs_dl_t *)malloc(sizeof(rs_dl_t));
  rs_dl_build(cell_n_id_cell(cell), cell->cp_type, R);
  trk_raw_t *raw = (trk_raw_t *)malloc(sizeof(trk_raw_t) * (n_sym + 1));
  trk_filt_t *fl = (trk_filt_t *)malloc(sizeof(trk_filt_t) * (n_sym + 1));
  for (int port = 0; port < 4; port++) { n_meas[port] = 0; ce_upto[port] = 0; }
  for (int port = 0; port < cell->n_ports; port++) {
    int m = 0, slot = slot0, sym = sym0;
    for (int i = 0; i < n_sym; i++) {                                 /* :868-890 */
      const double shift = rs_get_shift(R, slot, sym, port);
      if (!isnan(shift)) {
        trk_raw_t *r = &raw[m++];
        r->shift = shift; r->slot = slot; r->sym = sym; r->idx = i;
        r->frequency_offset = freq_off[i]; r->frame_timing = frame_timing[i];
        const cd *rs = rs_get_rs(R, slot, sym);
        for (int k = 0; k < 12; k++) r->ce[k] = cmul(syms[(size_t)i * 72 + round_i(shift) + 6 * k], cconj(rs[k]));
      }
      slot_sym_inc(n_symb_dl, &slot, &sym);
    }
    int nf = 0;
    for (int r = 1; r + 1 < m; r++) {
      const trk_raw_t *prev = &raw[r - 1], *curr = &raw[r], *next = &raw[r + 1];
      trk_filt_t *F = &fl[nf];
      trk_filter_ce(prev, curr, next, F->ce_filt);                    /* :906 */
      cd diff[12];
      for (int k = 0; k < 12; k++) diff[k] = csub(curr->ce[k], F->ce_filt[k]);
      const double np = sigpower(diff, 12) * 7 / 6;                   /* :908 */
      const double tp = sigpower(F->ce_filt, 12);
      const double sp_raw = tp - np / 7;
      const double sp = (.00001 > sp_raw) ? .00001 : sp_raw;          /* MAX(.00001, sp_raw) */
      F->shift = curr->shift; F->slot = curr->slot; F->sym = curr->sym; F->idx = curr->idx;
      F->tp = tp; F->sp = sp; F->sp_raw = sp_raw; F->np = np;
      /* do_foe :203-243 */
      cd foe_comb = c_(0, 0);
      double foe_comb_np = 0, wsum = 0;
      for (int k = 0; k < 12; k++) {
        const cd foe = cmul(cconj(prev->ce[k]), next->ce[k]);
        const double f2 = cabs2(F->ce_filt[k]);
        const double foe_np = np * np + 2 * np * f2;
        const double weight = f2 / foe_np;
        foe_comb = cadd(foe_comb, cscale(foe, weight));
        foe_comb_np += foe_np * weight * weight;
        wsum += f2 * weight;
      }
      const double scale = 1 / wsum;
      foe_comb = cscale(foe_comb, scale);
      foe_comb_np = foe_comb_np * scale * scale;
      const double frequency_offset = prev->frequency_offset;
      const double k_factor = (fc_requested - frequency_offset) / fc_programmed;
      const double residual_f = carg_(foe_comb) / (2 * PI) /
                                (0.0005 + WRAP(next->frame_timing - prev->frame_timing, -19200.0 / 2, 19200.0 / 2) * (1 / (fs_programmed * k_factor)));
      const double residual_f_np = (foe_comb_np / 2 > .001) ? foe_comb_np / 2 : .001;
      /* do_toe_v2 :245-288 */
      cd toe1 = c_(0, 0), toe2 = c_(0, 0);
      const trk_raw_t *A = (prev->shift < curr->shift) ? prev : curr, *B = (prev->shift < curr->shift) ? curr : prev;
      for (int k = 0; k < 12; k++) toe1 = cadd(toe1, cmul(cconj(A->ce[k]), B->ce[k]));
      toe1 = cdivr(toe1, 12);
      {
        cd s1 = c_(0, 0), s2 = c_(0, 0);
        for (int k = 0; k <= 4; k++) s1 = cadd(s1, cmul(cconj(B->ce[k]), A->ce[k + 1]));
        for (int k = 6; k <= 10; k++) s2 = cadd(s2, cmul(cconj(B->ce[k]), A->ce[k + 1]));
        toe2 = cdivr(cadd(s1, s2), 10);
      }
      toe1 = cdivr(toe1, sqrt(sp));
      toe2 = cdivr(toe2, sqrt(sp));
      const double delay = -(carg_(toe1) + carg_(toe2)) / 2 / 3 / (2 * PI / 128);
      const double delay_np = (np / sp / 2 / 12 > .001) ? np / sp / 2 / 12 : .001;
      if (nf < max_rs) {
        double *mrow = meas + ((size_t)port * max_rs + nf) * 9;
        mrow[0] = curr->idx; mrow[1] = np; mrow[2] = tp; mrow[3] = sp_raw; mrow[4] = sp;
        mrow[5] = frequency_offset + residual_f; mrow[6] = residual_f_np;
        mrow[7] = curr->frame_timing + delay; mrow[8] = delay_np;
      }
      nf++;
    }
    n_meas[port] = nf < max_rs ? nf : max_rs;
    if (dbg && dbg->port == port) {
      for (int r = 0; r < m; r++) { memcpy(dbg->raw + (size_t)r * 12, raw[r].ce, sizeof(cd) * 12); dbg->raw_idx[r] = raw[r].idx; }
      for (int r = 0; r < nf; r++) { memcpy(dbg->filt + (size_t)r * 12, fl[r].ce_filt, sizeof(cd) * 12); dbg->filt_idx[r] = fl[r].idx; }
      *dbg->n_raw = m; *dbg->n_filt = nf;
    }
    /* interp2d :402-477 over consecutive filtered RS symbols */
    cd *ce = (cd *)ce_re_im + (size_t)port * n_sym * 72;
    double *pw = ce_pw + (size_t)port * n_sym * 4;
    int initialized = 0;
    for (int j = 0; j + 1 < nf; j++) {
      const trk_filt_t *rp = &fl[j], *rc = &fl[j + 1];
      cd ip[72], ic[72];
      trk_interp72(rp, ip);
      trk_interp72(rc, ic);
      double time_diff;
      if (port > 2) time_diff = 0.0005;                               /* quirk kept: ports 0..2 take the symbol-based spacing */
      else if (cell->cp_type == ORC_CP_EXTENDED) time_diff = 3 * (128 + 32) * (1 / (FS_LTE / 16));
      else if (rp->sym == 0) time_diff = 4 * (128 + 9) * (1 / (FS_LTE / 16));
      else time_diff = (2 * (128 + 9) + (128 + 10)) * (1 / (FS_LTE / 16));
      double time_offset = 0;
      int sl = rp->slot, sy = rp->sym, i = rp->idx;
      while ((sl != rc->slot) || (sy != rc->sym)) {
        cd mid[72];
        const double q = time_offset / time_diff;
        for (int t = 0; t < 72; t++) mid[t] = cadd(ip[t], cscale(csub(ic[t], ip[t]), q));
        const double m_tp = rp->tp + (rc->tp - rp->tp) * q, m_sp = rp->sp + (rc->sp - rp->sp) * q;
        const double m_spr = rp->sp_raw + (rc->sp_raw - rp->sp_raw) * q, m_np = rp->np + (rc->np - rp->np) * q;
        if (!initialized) {                                           /* :449-462: the first estimate also covers the symbols before it */
          initialized = 1;
          for (int b = 0; b < i; b++) { memcpy(ce + (size_t)b * 72, mid, sizeof(mid)); pw[b * 4] = m_tp; pw[b * 4 + 1] = m_sp; pw[b * 4 + 2] = m_spr; pw[b * 4 + 3] = m_np; }
        }
        memcpy(ce + (size_t)i * 72, mid, sizeof(mid));
        pw[i * 4] = m_tp; pw[i * 4 + 1] = m_sp; pw[i * 4 + 2] = m_spr; pw[i * 4 + 3] = m_np;
        if (cell->cp_type == ORC_CP_EXTENDED) time_offset += (128 + 32) * (1 / (FS_LTE / 16));
        else if (sy == 6) time_offset += (128 + 10) * (1 / (FS_LTE / 16));
        else time_offset += (128 + 9) * (1 / (FS_LTE / 16));
        slot_sym_inc(n_symb_dl, &sl, &sy);
        i++;
      }
      ce_upto[port] = i;
    }
  }
  free(R); free(raw); free(fl);
  return 0;
}
int orc_trk_chan_est(const orc_cell *cell, const double *syms_re_im, int n_sym, int slot0, int sym0, const double *freq_off,
                     const double *frame_timing, double fc_requested, double fc_programmed, double fs_programmed,
                     double *meas, int max_rs, int *n_meas, double *ce_re_im, double *ce_pw, int *ce_upto) {
  return trk_chan_est_x(cell, syms_re_im, n_sym, slot0, sym0, freq_off, frame_timing, fc_requested, fc_programmed, fs_programmed,
                        meas, max_rs, n_meas, ce_re_im, ce_pw, ce_upto, NULL);
}
/* The tracker's raw reference-signal estimates (ref src/tracker_thread.cpp:868-890) and filter_ce outputs (:176-201) of one
 * port, [n][12] each with the symbol index of every row -- exported so that the tests can pin them to the searcher's
 * chan_est (orc_chan_est_dbg) on the same grid. */
int orc_trk_raw_filt(const orc_cell *cell, const double *syms_re_im, int n_sym, int slot0, int sym0, int port,
                     double *raw_re_im, int *raw_idx, int *n_raw, double *filt_re_im, int *filt_idx, int *n_filt) {
  double *meas = (double *)malloc(sizeof(double) * 4 * (size_t)n_sym * 9), *fo = (double *)calloc(n_sym, sizeof(double));
  cd *ce = (cd *)malloc(sizeof(cd) * 4 * (size_t)n_sym * 72);
  double *pw = (double *)malloc(sizeof(double) * 4 * (size_t)n_sym * 4);
  int nm[4], upto[4];
  trk_dbg_t d = {port, (cd *)raw_re_im, (cd *)filt_re_im, raw_idx, filt_idx, n_raw, n_filt};
  *n_raw = *n_filt = 0;
  const int rc = trk_chan_est_x(cell, syms_re_im, n_sym, slot0, sym0, fo, fo, 739e6, 739e6, 1.92e6, meas, n_sym, nm, (double *)ce, pw,
                                upto, &d);
  free(meas); free(fo); free(ce); free(pw);
  return rc;
}

/* ref: src/tracker_thread.cpp:318-341 do_ac_fd, :343-371 do_ac_td, :754-820 do_pss_sss_sigpower_ce -- the display
 * statistics the tracker thread derives from the same per-symbol data: the frequency-domain autocorrelation of the
 * raw reference-signal estimates of the current RS symbol (normalised by its signal power), the time-domain
 * autocorrelation against the 71 RS symbols before it, and signal / noise power + a smoothed channel estimate from
 * every PSS/SSS pair.  Here WITHOUT the running averages they feed (tracked_cell.ac_fd / ac_td / sync_*_av: scalar
 * recurrences, kept by the caller like the other feedback loops).  `meas` / `n_meas` are orc_trk_chan_est's outputs
 * of the same block (row f = filtered RS symbol f + 1 of the port; column 4 = rs_curr_sp).
 *   ac_fd   [4][max_rs][12] complex: row f = do_ac_fd's ac_fd before the weighting (:324-334)
 *   ac_td   [4][max_rs][72] complex: row f = do_ac_td's this_xc (:361-364), defined for f >= 71 (NaN before)
 *   sync    [max_hf][4] = tp, sp, np, np_blank of the k-th PSS/SSS pair of the block; sync_ce [max_hf][72] complex */
int orc_trk_stats(const orc_cell *cell, const double *syms_re_im, int n_sym, int slot0, int sym0, const double *meas, int max_rs,
                  const int *n_meas, double *ac_fd, double *ac_td, double *sync, double *sync_ce, int max_hf, int *n_hf) {
  const int n_symb_dl = cell_n_symb_dl(cell);
  if (n_symb_dl < 0 || cell->n_ports < 1 || cell->n_ports > 4) return -1;
  const cd *syms = (const cd *)syms_re_im;
  rs_dl_t *R = (rs_dl_t *)malloc(sizeof(rs_dl_t));
  rs_dl_build(cell_n_id_cell(cell), cell->cp_type, R);
  cd (*raw)[12] = (cd (*)[12])malloc(sizeof(cd) * 12 * (n_sym + 1));
  for (int port = 0; port < cell->n_ports; port++) {
    int m = 0, slot = slot0, sym = sym0;
    for (int i = 0; i < n_sym; i++) {                                 /* rs_curr.ce as in the main loop, :868-890 */
      const double shift = rs_get_shift(R, slot, sym, port);
      if (!isnan(shift)) {
        const cd *rs = rs_get_rs(R, slot, sym);
        for (int k = 0; k < 12; k++) raw[m][k] = cmul(syms[(size_t)i * 72 + round_i(shift) + 6 * k], cconj(rs[k]));
        m++;
      }
      slot_sym_inc(n_symb_dl, &slot, &sym);
    }
    const int nf = (n_meas[port] < max_rs) ? n_meas[port] : max_rs;
    for (int f = 0; f < nf; f++) {
      const cd *cur = raw[f + 1];
      const double sp = meas[((size_t)port * max_rs + f) * 9 + 4];
      if (ac_fd) {
        cd *o = (cd *)ac_fd + ((size_t)port * max_rs + f) * 12;
        for (int d = 0; d < 12; d++) {                                /* :325-334 */
          cd a = c_(0, 0);
          for (int t = 0; t < 12 - d; t++) a = cadd(a, cmul(cconj(cur[t]), cur[t + d]));
          a = cdivr(a, 12 - d);
          o[d] = cdivr(a, sp);
        }
      }
      if (ac_td) {
        cd *o = (cd *)ac_td + ((size_t)port * max_rs + f) * 72;
        for (int t = 0; t < 72; t++) {
          if (f < 71) { o[t] = c_(NAN, NAN); continue; }              /* the 72-deep history is not full yet, :358 */
          const cd *old = raw[f + 1 - t];
          cd a = c_(0, 0);
          for (int k = 0; k < 12; k++) a = cadd(a, cmul(cconj(cur[k]), old[k]));     /* elem_mult_sum(conj(h[71]), h[71-t]) */
          o[t] = cdivr(cdivr(a, 12), sp);
        }
      }
    }
  }
  /* do_pss_sss_sigpower_ce, :754-820 */
  int hf = 0;
  {
    static cd pss_fd_tab[3][62]; static int ready = 0;
#pragma omp critical(orc_trk_stats_tab)
    if (!ready) { for (int t = 0; t < 3; t++) pss_fd_calc(t, pss_fd_tab[t]); ready = 1; }
    const cd *sss_sym = NULL;
    int slot = slot0, sym = sym0;
    for (int i = 0; i < n_sym; i++) {
      if ((slot == 0 || slot == 10) && sym == n_symb_dl - 2) sss_sym = syms + (size_t)i * 72;
      else if ((slot == 0 || slot == 10) && sym == n_symb_dl - 1 && sss_sym) {
        const cd *pss_sym = syms + (size_t)i * 72;
        const double np_blank = (sigpower(sss_sym, 5) + sigpower(sss_sym + 67, 5) + sigpower(pss_sym, 5) + sigpower(pss_sym + 67, 5)) / 4;
        int32_t sfd[62];
        sss_fd_calc(cell->n_id_1, cell->n_id_2, (slot == 0) ? 0 : 10, sfd);
        cd ce_sss[62], ce_pss[62], sm[62], d1[62], d2[62];
        for (int t = 0; t < 62; t++) {
          ce_sss[t] = cmul(sss_sym[5 + t], c_((double)sfd[t], 0));
          ce_pss[t] = cmul(pss_sym[5 + t], cconj(pss_fd_tab[cell->n_id_2][t]));
        }
        for (int t = 0; t < 62; t++) {
          const int lt = (t - 6 > 0) ? t - 6 : 0, rt = (t + 6 < 61) ? t + 6 : 61;
          cd a = c_(0, 0), b = c_(0, 0);
          for (int q = lt; q <= rt; q++) a = cadd(a, ce_sss[q]);
          for (int q = lt; q <= rt; q++) b = cadd(b, ce_pss[q]);
          sm[t] = cdivr(cadd(a, b), 2 * (rt - lt + 1));
        }
        for (int t = 0; t < 62; t++) { d1[t] = csub(sm[t], ce_sss[t]); d2[t] = csub(sm[t], ce_pss[t]); }
        const double np = (sigpower(d1, 62) * 13 / 12 + sigpower(d2, 62) * 13 / 12) / 2;
        const double tp = sigpower(sm, 62);
        const double sp = tp - np / 13;
        if (hf < max_hf) {
          if (sync) { double *o = sync + (size_t)hf * 4; o[0] = tp; o[1] = sp; o[2] = np; o[3] = np_blank; }
          if (sync_ce) {
            cd *o = (cd *)sync_ce + (size_t)hf * 72;
            for (int t = 0; t < 72; t++) o[t] = (t >= 5 && t < 67) ? sm[t - 5] : c_(0, 0);
          }
        }
        hf++;
      }
      slot_sym_inc(n_symb_dl, &slot, &sym);
    }
  }
  if (n_hf) *n_hf = hf;
  free(raw); free(R);
  return 0;
}

/* One MIB attempt of the tracker: pbch_extract_rt (ref src/tracker_thread.cpp:494-529) + the decode part of
 * do_mib_decode (:555-705) on 16 PBCH symbols (slot 1, symbols 0..3 of four consecutive frames).
 * syms16 [16][72]; ce16 [n_ports][16][72]; np16 [n_ports][16].  c_est: the 40 decoded bits; *crc_ok: CRC (with the
 * port mask) matches; *fields_ok: bandwidth / PHICH fields equal the tracked cell's (the lock test of :689-694 is
 * crc_ok && fields_ok). */
int orc_trk_mib(const orc_cell *cell, const double *syms16, const double *ce16, const double *np16, uint8_t *c_est,
                int *crc_ok, int *fields_ok) {
  const int n_ports = cell->n_ports;
  if (n_ports != 1 && n_ports != 2 && n_ports != 4) return -1;
  const int m_bit = (cell->cp_type == ORC_CP_NORMAL) ? 1920 : 1728, n_syms = m_bit / 2;
  const int v_shift_m3 = imod(cell_n_id_cell(cell), 3);
  const cd *S = (const cd *)syms16, *CE = (const cd *)ce16;
  cd *pbch_sym = (cd *)malloc(sizeof(cd) * n_syms), *pbch_ce = (cd *)malloc(sizeof(cd) * 4 * n_syms);
  double *np_pre = (double *)malloc(sizeof(double) * 4 * n_syms);
  cd *syms_mib = (cd *)malloc(sizeof(cd) * n_syms);
  double *np_mib = (double *)malloc(sizeof(double) * n_syms), *e_est = (double *)malloc(sizeof(double) * m_bit);
  uint8_t *scr = (uint8_t *)malloc(m_bit);
  int idx = 0;
  for (int fr = 0; fr < 4; fr++) for (int symn = 0; symn < 4; symn++) for (int sc = 0; sc < 72; sc++) {
    if ((imod(sc, 3) == v_shift_m3) && ((symn == 0) || (symn == 1) || ((symn == 3) && (cell->cp_type == ORC_CP_EXTENDED)))) continue;
    pbch_sym[idx] = S[(fr * 4 + symn) * 72 + sc];
    for (int p = 0; p < n_ports; p++) { pbch_ce[p * n_syms + idx] = CE[((size_t)p * 16 + fr * 4 + symn) * 72 + sc]; np_pre[p * n_syms + idx] = np16[p * 16 + fr * 4 + symn]; }
    idx++;
  }
  if (n_ports == 1) {                                                  /* :572-576 */
    for (int t = 0; t < n_syms; t++) {
      const cd h = pbch_ce[t];
      const cd gain = cconj(cdiv(h, c_(cabs2(h), 0)));
      syms_mib[t] = cmul(pbch_sym[t], gain);
      np_mib[t] = np_pre[t] * cabs2(gain);
    }
  } else {                                                             /* :584-625 */
    for (int t = 0; t < n_syms; t += 2) {
      cd h1, h2; double np_temp;
      if (n_ports == 2) {
        h1 = cdivr(cadd(pbch_ce[t], pbch_ce[t + 1]), 2);
        h2 = cdivr(cadd(pbch_ce[n_syms + t], pbch_ce[n_syms + t + 1]), 2);
        np_temp = (np_pre[t] + np_pre[n_syms + t]) / 2;
      } else if (imod(t, 4) == 0) {
        h1 = cdivr(cadd(pbch_ce[t], pbch_ce[t + 1]), 2);
        h2 = cdivr(cadd(pbch_ce[2 * n_syms + t], pbch_ce[2 * n_syms + t + 1]), 2);
        np_temp = (np_pre[t] + np_pre[2 * n_syms + t]) / 2;
      } else {
        h1 = cdivr(cadd(pbch_ce[n_syms + t], pbch_ce[n_syms + t + 1]), 2);
        h2 = cdivr(cadd(pbch_ce[3 * n_syms + t], pbch_ce[3 * n_syms + t + 1]), 2);
        np_temp = (np_pre[n_syms + t] + np_pre[3 * n_syms + t]) / 2;
      }
      const cd x1 = pbch_sym[t], x2 = pbch_sym[t + 1];
      const double scale = pow(h1.re, 2) + pow(h1.im, 2) + pow(h2.re, 2) + pow(h2.im, 2);
      syms_mib[t] = cdivr(cadd(cmul(cconj(h1), x1), cmul(h2, cconj(x2))), scale);
      syms_mib[t + 1] = cconj(cdivr(cadd(cmul(c_(-h2.re, h2.im), x1), cmul(h1, cconj(x2))), scale));
      np_mib[t] = (pow(hypot(h1.re, h1.im) / scale, 2) + pow(hypot(h2.re, h2.im) / scale, 2)) * np_temp;
      np_mib[t + 1] = np_mib[t];
    }
    const double s2 = pow(2, 0.5);
    for (int t = 0; t < n_syms; t++) syms_mib[t] = cscale(syms_mib[t], s2);
  }
  lte_demodulate_qpsk(syms_mib, np_mib, n_syms, e_est);               /* :634 */
  orc_lte_pn((uint32_t)cell_n_id_cell(cell), (uint32_t)m_bit, scr);
  for (int t = 0; t < m_bit; t++) if (scr[t]) e_est[t] = -e_est[t];
  double d_est[3 * 40];
  lte_conv_deratematch(e_est, m_bit, 40, d_est);
  uint8_t crc_est[16];
  conv_decode_tailbite(d_est, 40, c_est);
  crc16_bits(c_est, 24, crc_est);
  if (n_ports == 2) for (int t = 0; t < 16; t++) crc_est[t] = 1 - crc_est[t];
  else if (n_ports == 4) for (int t = 1; t < 16; t += 2) crc_est[t] = 1 - crc_est[t];
  *crc_ok = memcmp(crc_est, c_est + 24, 16) == 0;
  const int bw_packed = c_est[0] * 4 + c_est[1] * 2 + c_est[2];
  static const int bw[8] = {6, 15, 25, 50, 75, 100, 0, 0};
  const int n_rb_dl_est = bw[bw_packed];
  const int phich_duration_est = c_est[3] ? 2 : 1;
  const int phich_resource_est = 1 + c_est[4] * 2 + c_est[5];
  *fields_ok = (n_rb_dl_est == cell->n_rb_dl) && (phich_duration_est == cell->phich_duration) && (phich_resource_est == cell->phich_resource);
  free(pbch_sym); free(pbch_ce); free(np_pre); free(syms_mib); free(np_mib); free(e_est); free(scr);
  return 0;
}

/* ---- thin exports of two third-party-arithmetic restatements, so that the tests can pin them to independent
 * implementations (numpy.linalg.solve; the closed form of the QPSK log-MAP LLR) ---- */
/* ---- the producer thread's sample loop for ONE tracked cell (ref src/producer_thread.cpp:96-131, 196-246) ---------------------
 * Restated as written there: the timestamp is ACCUMULATED -- sample_time += (FS_LTE/16) / (fs_programmed k_factor), minus 19200 once it
 * exceeds 19200 (:127-131; the reference starts at -1, here the caller gives the timestamp of the buffer's first sample and the
 * accumulator starts one step before it) -- and every sample is examined in turn (:200): while not filling, a capture starts when
 * tdiff = WRAP(timestamp - (frame_timing + target_cap_start_time), -9600, 9600) satisfies |tdiff| < 0.5 or 0 < tdiff < 3 (:203-213),
 * with late = tdiff (:216); 128 samples later (:225) the target advances by 32 + 128 (extended CP), 128 + 10 (after symbol 6) or
 * 128 + 9, modulo 19200 (:236-241), and slot_sym_inc steps the symbol number (:242).  frame_timing and the frequency offset are held
 * (a recorded buffer: the values at the beginning of the capture, :219-220).  A capture the buffer's end cuts off is not reported.
 * hit[k] = index of the first sample of symbol k's capture.  Returns the number of complete captures. */
int orc_producer_cut(uint32_t n_cap, double ts_first, double frame_timing, int cp_type, double frequency_offset, double fc_requested,
                     double fc_programmed, double fs_programmed, int n_sym_max, int32_t *hit, double *late) {
  const double k_factor = (fc_requested - frequency_offset) / fc_programmed;      /* :99 */
  const double step = (FS_LTE / 16) / (fs_programmed * k_factor);                  /* :127 */
  double sample_time = ts_first - step;
  double target_cap_start_time = (cp_type == ORC_CP_NORMAL) ? 10 : 32;             /* :172 */
  int sym_num = 0, filling = 0, buffer_offset = 0, n = 0;
  const int n_symb_dl = (cp_type == ORC_CP_NORMAL) ? 7 : 6;
  for (uint32_t t = 0; t < n_cap && n < n_sym_max; ++t) {
    sample_time += step;                                                           /* :127 */
    if (sample_time > 19200.0) sample_time -= 19200.0;                             /* :129-130 */
    if (!filling) {                                                                /* :202 */
      const double tdiff = WRAP(sample_time - (frame_timing + target_cap_start_time), -19200.0 / 2, 19200.0 / 2);
      if (fabs(tdiff) < 0.5 || (tdiff > 0 && tdiff < 3)) {                         /* :204-210 */
        filling = 1;
        buffer_offset = 0;
        hit[n] = (int32_t)t;
        late[n] = tdiff;                                                           /* :216 */
      }
    }
    if (filling) {                                                                 /* :225 */
      if (++buffer_offset == 128) {                                                /* :227 */
        ++n;
        filling = 0;
        if (cp_type == ORC_CP_EXTENDED) target_cap_start_time += 32 + 128;         /* :237 */
        else target_cap_start_time += (sym_num == 6) ? 128 + 10 : 128 + 9;         /* :239 */
        target_cap_start_time = fmod(target_cap_start_time, 19200);                /* :241 (itpp mod on a non-negative value) */
        sym_num = (sym_num + 1) % n_symb_dl;                                       /* :242 slot_sym_inc */
      }
    }
  }
  return n;
}

void orc_solve3(const double *M_re_im /*[3][3]*/, const double *V_re_im /*[3]*/, double *out_re_im /*[3]*/) {
  cd M[3][3], V[3], o[3];
  memcpy(M, M_re_im, sizeof(M));
  memcpy(V, V_re_im, sizeof(V));
  solve3(M, V, o);
  memcpy(out_re_im, o, sizeof(o));
}
void orc_qpsk_llr(const double *syms_re_im, const double *np, int n, double *llr /*[2n]*/) {
  lte_demodulate_qpsk((const cd *)syms_re_im, np, n, llr);
}
