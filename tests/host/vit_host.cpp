// Host-side harness for the two-pass tail-biting Viterbi of lte_device.h (the same __host__ __device__ code the GPU
// runs, with the 64 lanes walked sequentially): tests/test_viterbi_host.py compares it with the oracle's
// exhaustive tail-biting decoder on random, quantised (tie-prone) and saturated inputs.  Test infrastructure.
#include "../../lte-cell-scanner_amd/csrc/lte_device.h"
#include <cstring>
#include <vector>

extern "C" int vit_host_decode(const double *d_est /*[3][40]*/, unsigned long long *bits40, int *best_ss, double *best_metric) {
  double best = INFINITY;
  int bss = -1;
  bool finite = true;                                  // the device's choice of the step's form (pbch_decode_wave)
  for (int i = 0; i < 120; ++i) finite = finite && vit_finite(d_est[i]);
  for (int ss = 0; ss < 64; ++ss) {                    // pass 1: end metrics, no survivors
    const double fin = finite ? vit_end_metric<true>(d_est, d_est + 40, d_est + 80, ss) : vit_end_metric<false>(d_est, d_est + 40, d_est + 80, ss);
    if (fin < best) { best = fin; bss = ss; }
  }
  *best_ss = bss;
  *best_metric = best;
  double again = 0;
  *bits40 = (bss >= 0) ? vit_retrace_host(d_est, d_est + 40, d_est + 80, bss, &again) : 0ull;      // pass 2: the winner, one state per lane
  if (bss >= 0 && !(again == best)) return -1;         // the two passes form the same sums: the same end metric, bit for bit
  return bss >= 0;
}
// both forms of the step on the same input (finite inputs: they must agree bit for bit), and for every start state the retrace's end
// metric must be pass 1's
extern "C" int vit_host_forms_agree(const double *d_est /*[3][40]*/) {
  for (int ss = 0; ss < 64; ++ss) {
    const double fa = vit_end_metric<true>(d_est, d_est + 40, d_est + 80, ss);
    const double fb = vit_end_metric<false>(d_est, d_est + 40, d_est + 80, ss);
    double fc = 0;
    (void)vit_retrace_host(d_est, d_est + 40, d_est + 80, ss, &fc);
    if (!(fa == fb) || !(fa == fc)) return 0;
  }
  return 1;
}
extern "C" int vit_host_crc_ok(unsigned long long bits, int n_ports) { return pbch_crc_ok(bits, n_ports); }

// ---- the register-resident transform's building blocks (lte_device.h: fft16, fft8, cis_small), the same __host__ __device__ code the
// kernels run; tests/test_viterbi_host.py compares them with numpy's FFT and libm
extern "C" void fft_host_16(const double *in_re_im, double *out_re_im) {
  cd2 x[16];
  for (int j = 0; j < 16; ++j) x[j] = mk(in_re_im[2 * j], in_re_im[2 * j + 1]);
  fft16(x);
  for (int k = 0; k < 16; ++k) { const cd2 v = fft16_out(x, k); out_re_im[2 * k] = v.re; out_re_im[2 * k + 1] = v.im; }
}
extern "C" void fft_host_8(const double *in_re_im, double *out_re_im) {
  cd2 x[8];
  for (int j = 0; j < 8; ++j) x[j] = mk(in_re_im[2 * j], in_re_im[2 * j + 1]);
  fft8(x);
  for (int k = 0; k < 8; ++k) { out_re_im[2 * k] = x[k].re; out_re_im[2 * k + 1] = x[k].im; }
}
// the whole 128-point transform in the kernels' decomposition (lane l holds points l + 8 j; 16-point transforms, twiddles W128^(l k2),
// 8-point transforms over l: X[k2 + 16 k1]) -- the data movement of fft128_x8 written as plain loops
extern "C" void fft_host_128(const double *in_re_im, double *out_re_im) {
  cd2 z[8][16];
  for (int l = 0; l < 8; ++l) {
    cd2 x[16];
    for (int j = 0; j < 16; ++j) x[j] = mk(in_re_im[2 * (l + 8 * j)], in_re_im[2 * (l + 8 * j) + 1]);
    fft16(x);
    for (int k2 = 0; k2 < 16; ++k2) {
      const double a = -M_PI * (double)(l * k2) / 64.0;
      z[l][k2] = k2 ? cmul(fft16_out(x, k2), mk(cos(a), sin(a))) : fft16_out(x, k2);
    }
  }
  for (int k2 = 0; k2 < 16; ++k2) {
    cd2 y[8];
    for (int l = 0; l < 8; ++l) y[l] = z[l][k2];
    fft8(y);
    for (int k1 = 0; k1 < 8; ++k1) { out_re_im[2 * (k2 + 16 * k1)] = y[k1].re; out_re_im[2 * (k2 + 16 * k1) + 1] = y[k1].im; }
  }
}
extern "C" void cis_small_host(double x, double *out) { const cd2 v = cis_small(x); out[0] = v.re; out[1] = v.im; }

// the phase walk's WRAP without its division (lte_device.h: trk_wrap_certain_interval): for every x and every candidate floor
// around WRAP's own, a k = x + pi inside the candidate's interval must have that candidate as WRAP's floor, and the walk's value
// (k - nf) - pi must be WRAP's bit for bit.  Returns the violations; *n_inside = how often WRAP's own floor was accepted.
extern "C" long trk_wrap_interval_violations(const double *x, long n, long *n_inside) {
  long bad = 0;
  *n_inside = 0;
  const double wn = M_PI - (-M_PI);
  for (long i = 0; i < n; ++i) {
    const double k = x[i] - (-M_PI);
    const double q = floor(k / wn);
    const double want = trk_wrap(x[i], -M_PI, M_PI);
    for (int d = -2; d <= 2; ++d) {
      double nf, lo, hi;
      trk_wrap_certain_interval(q + d, nf, lo, hi);
      if (!(k >= lo && k < hi)) continue;
      const double got = (k - nf) + (-M_PI);
      unsigned long long ua, ub;
      memcpy(&ua, &want, 8); memcpy(&ub, &got, 8);
      if (d != 0 || ua != ub) ++bad;
      if (d == 0) ++*n_inside;
    }
  }
  return bad;
}
