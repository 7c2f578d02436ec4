// Host-side harness for the lane-per-trellis Viterbi of lte_device.h (the same __host__ __device__ code the GPU
// runs, with the 64 lanes walked sequentially): tests/test_viterbi_host.py compares it with the oracle's
// exhaustive tail-biting decoder on random, quantised (tie-prone) and saturated inputs.  Test infrastructure.
#include "../../lte-cell-scanner_amd/csrc/lte_device.h"
#include <vector>

extern "C" int vit_host_decode(const double *d_est /*[3][40]*/, unsigned long long *bits40, int *best_ss, double *best_metric) {
  std::vector<unsigned long long> surv((size_t)40 * 64);
  double best = INFINITY;
  int bss = -1;
  bool finite = true;                                  // the device's choice of the step's form (pbch_decode_wave)
  for (int i = 0; i < 120; ++i) finite = finite && vit_finite(d_est[i]);
  for (int ss = 0; ss < 64; ++ss) {
    const double fin = finite ? vit_trellis<true>(d_est, d_est + 40, d_est + 80, ss, surv.data() + ss, 64)
                              : vit_trellis<false>(d_est, d_est + 40, d_est + 80, ss, surv.data() + ss, 64);
    if (fin < best) { best = fin; bss = ss; }
  }
  *best_ss = bss;
  *best_metric = best;
  *bits40 = (bss >= 0) ? vit_traceback(surv.data() + bss, 64, bss) : 0ull;
  return bss >= 0;
}
// both forms of the step on the same input (finite inputs: they must agree bit for bit, survivor words included)
extern "C" int vit_host_forms_agree(const double *d_est /*[3][40]*/) {
  std::vector<unsigned long long> a((size_t)40 * 64), b((size_t)40 * 64);
  for (int ss = 0; ss < 64; ++ss) {
    const double fa = vit_trellis<true>(d_est, d_est + 40, d_est + 80, ss, a.data() + ss, 64);
    const double fb = vit_trellis<false>(d_est, d_est + 40, d_est + 80, ss, b.data() + ss, 64);
    if (!(fa == fb)) return 0;
  }
  return a == b;
}
extern "C" int vit_host_crc_ok(unsigned long long bits, int n_ports) { return pbch_crc_ok(bits, n_ports); }
