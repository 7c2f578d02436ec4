// lte_tables.cpp -- host-side constant tables of the product (uploaded once per context).
//
// Independent derivation from 3GPP TS 36.211 of what the reference builds in
// src/lte_lib.cpp (PSS_fd :155-174, PSS_td :177-188, SSS_fd :199-274, lte_pn :41-147) and
// of chi2cdf_inv (include/dsp.h:188-193).  tests/test_tables.py checks every table
// against the CPU oracle's restatement of the reference code.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "lcs_internal.h"

namespace lcs_tables {

static const double kPi = 3.14159265358979323846;

// d_u(n) = exp(-j*pi*u*n*(n+1)/63), n = 0..62 without the DC element n = 31 (36.211 6.11.1.1)
void pss_fd(int n_id_2, double *re_im) {
  static const int root[3] = {25, 29, 34};
  int o = 0;
  for (int n = 0; n < 63; ++n) {
    if (n == 31) continue;
    // same operation order as the reference's exp((j*-pi*u/63) * (n*(n+1))): the phase reaches
    // ~4.9e3 rad, so the rounding of this product is visible at the 1e-12 level
    const double ph = ((-kPi * root[n_id_2]) / 63.0) * (double)(n * (n + 1));
    re_im[2 * o] = std::cos(ph);
    re_im[2 * o + 1] = std::sin(ph);
    ++o;
  }
}

// 128-point OFDM symbol of the PSS scaled so that the 62 occupied carriers give unit sample
// power (x sqrt(128)*sqrt(128/62)/128), with the 9-sample cyclic prefix in front.
void pss_td(int n_id_2, double *re_im) {
  double fd[62 * 2];
  pss_fd(n_id_2, fd);
  double X[128][2];
  std::memset(X, 0, sizeof(X));
  for (int i = 0; i < 31; ++i) {
    X[1 + i][0] = fd[2 * (31 + i)]; X[1 + i][1] = fd[2 * (31 + i) + 1];   // positive carriers 1..31
    X[97 + i][0] = fd[2 * i];       X[97 + i][1] = fd[2 * i + 1];         // negative carriers -31..-1
  }
  const double scale = std::sqrt(128.0) * std::sqrt(128.0 / 62.0) / 128.0;
  double td[128][2];
  for (int n = 0; n < 128; ++n) {
    double sr = 0, si = 0;
    for (int k = 0; k < 128; ++k) {
      if (X[k][0] == 0 && X[k][1] == 0) continue;
      const int kn = (k * n) & 127;
      const double c = std::cos(2 * kPi * kn / 128.0), s = std::sin(2 * kPi * kn / 128.0);
      sr += X[k][0] * c - X[k][1] * s;
      si += X[k][0] * s + X[k][1] * c;
    }
    td[n][0] = sr * scale;
    td[n][1] = si * scale;
  }
  for (int i = 0; i < 9; ++i) { re_im[2 * i] = td[119 + i][0]; re_im[2 * i + 1] = td[119 + i][1]; }
  for (int i = 0; i < 128; ++i) { re_im[2 * (9 + i)] = td[i][0]; re_im[2 * (9 + i) + 1] = td[i][1]; }
}

// 36.211 6.11.2.1: m-sequences from their recursions, (m0,m1) from N_ID^(1).
void sss_fd(int n_id_1, int n_id_2, int slot_num, int32_t *out) {
  int xs[31] = {0, 0, 0, 0, 1}, xc[31] = {0, 0, 0, 0, 1}, xz[31] = {0, 0, 0, 0, 1};
  for (int i = 0; i < 26; ++i) {
    xs[i + 5] = (xs[i + 2] + xs[i]) & 1;
    xc[i + 5] = (xc[i + 3] + xc[i]) & 1;
    xz[i + 5] = (xz[i + 4] + xz[i + 2] + xz[i + 1] + xz[i]) & 1;
  }
  const int qp = n_id_1 / 30;
  const int q = (n_id_1 + qp * (qp + 1) / 2) / 30;
  const int mp = n_id_1 + q * (q + 1) / 2;
  const int m0 = mp % 31, m1 = (m0 + mp / 31 + 1) % 31;
  auto s = [&](int m, int n) { return 1 - 2 * xs[(n + m) % 31]; };
  auto c = [&](int off, int n) { return 1 - 2 * xc[(n + n_id_2 + off) % 31]; };
  auto z = [&](int m, int n) { return 1 - 2 * xz[(n + (m % 8)) % 31]; };
  for (int n = 0; n < 31; ++n) {
    int even, odd;
    if (slot_num == 0) { even = s(m0, n) * c(0, n); odd = s(m1, n) * c(3, n) * z(m0, n); }
    else               { even = s(m1, n) * c(0, n); odd = s(m0, n) * c(3, n) * z(m1, n); }
    out[2 * n] = even;
    out[2 * n + 1] = odd;
  }
}

// 36.211 7.2 length-31 Gold sequence, Nc = 1600, with both LFSRs held in 32-bit words.
void lte_pn(uint32_t c_init, uint32_t len, uint8_t *out) {
  uint32_t x1 = 1, x2 = c_init & 0x7fffffffu;
  for (uint32_t i = 0; i < 1600 + len; ++i) {
    if (i >= 1600) out[i - 1600] = (uint8_t)((x1 ^ x2) & 1u);
    const uint32_t n1 = ((x1 >> 3) ^ x1) & 1u;
    const uint32_t n2 = ((x2 >> 3) ^ (x2 >> 2) ^ (x2 >> 1) ^ x2) & 1u;
    x1 = (x1 >> 1) | (n1 << 30);
    x2 = (x2 >> 1) | (n2 << 30);
  }
}

// Jump-ahead table for the Gold generator above.  Both LFSRs are linear over GF(2), so the x2
// register after `steps` clocks is the XOR of out[b] over the set bits b of its initial value
// (out[b] = register reached from the unit state 1 << b); out[31] is the x1 register after the
// same number of clocks from its fixed initial state.  The CRS builder (k_cell_prep) uses it to
// reach c(2 * 104) of 36.211 6.10.1.1 in 31 XORs instead of 1808 clocks.
void pn_jump_table(uint32_t steps, uint32_t out[32]) {
  for (int b = 0; b < 32; ++b) {
    uint32_t x = (b < 31) ? (1u << b) : 1u;
    for (uint32_t i = 0; i < steps; ++i) {
      const uint32_t n = (b < 31) ? (((x >> 3) ^ (x >> 2) ^ (x >> 1) ^ x) & 1u) : (((x >> 3) ^ x) & 1u);
      x = (x >> 1) | (n << 30);
    }
    out[b] = x;
  }
}

// Inverse chi-square CDF: x with P(k/2, x/2) = p.  Regularised incomplete gamma through the
// Legendre continued fraction of the upper tail (all uses have p close to 1), solved with
// a bracketed secant/bisection in log-space of the tail probability.
static double upper_gamma_q(double a, double x) {
  if (x <= 0) return 1.0;
  if (x < a + 1) {   // series for P, return 1 - P
    double term = 1.0 / a, sum = term, ap = a;
    for (int n = 0; n < 10000; ++n) {
      ap += 1.0;
      term *= x / ap;
      sum += term;
      if (std::fabs(term) < std::fabs(sum) * 1e-17) break;
    }
    return 1.0 - sum * std::exp(-x + a * std::log(x) - std::lgamma(a));
  }
  const double tiny = 1e-300;
  double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
  for (int i = 1; i < 10000; ++i) {
    const double an = -i * (i - a);
    b += 2.0;
    d = an * d + b; if (std::fabs(d) < tiny) d = tiny;
    c = b + an / c; if (std::fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    const double del = d * c;
    h *= del;
    if (std::fabs(del - 1.0) < 1e-17) break;
  }
  return std::exp(-x + a * std::log(x) - std::lgamma(a)) * h;
}

double chi2cdf_inv(double p, double k) {
  const double a = k / 2.0, q = 1.0 - p;
  double lo = 0.0, hi = a + 10.0 * std::sqrt(a) + 50.0;
  while (upper_gamma_q(a, hi) > q) hi *= 2.0;
  for (int it = 0; it < 300; ++it) {
    const double mid = 0.5 * (lo + hi);
    if (mid == lo || mid == hi) break;
    if (upper_gamma_q(a, mid) > q) lo = mid; else hi = mid;
  }
  return lo + hi;   // 2 * midpoint
}

// Which coded bit (stream r in 0..2, column c in 0..39 -> r*40+c) each of the n_e rate-matched
// PBCH bits carries: 36.212 5.1.4.2 sub-block interleaver (32 columns, the SAME permutation for
// all three streams, <NULL> padding in front), bit collection stream after stream, circular
// selection skipping <NULL>s.
void pbch_deratematch_map(int n_e, uint8_t *out) {
  static const int perm[32] = {1, 17, 9, 25, 5, 21, 13, 29, 3, 19, 11, 27, 7, 23, 15, 31,
                               0, 16, 8, 24, 4, 20, 12, 28, 2, 18, 10, 26, 6, 22, 14, 30};
  const int D = 40, C = 32, R = (D + C - 1) / C, K = R * C, ND = K - D;
  std::vector<int> w(3 * K);
  for (int s = 0; s < 3; ++s)
    for (int col = 0; col < C; ++col)
      for (int row = 0; row < R; ++row) {
        const int y = row * C + perm[col];                 // position in the padded input
        w[s * K + col * R + row] = (y >= ND) ? s * D + (y - ND) : -1;
      }
  int k = 0, j = 0;
  while (k < n_e) {
    if (w[j] >= 0) out[k++] = (uint8_t)w[j];
    j = (j + 1) % (3 * K);
  }
}

}  // namespace lcs_tables
