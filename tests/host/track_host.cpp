// Host twin check of the tracker recurrences in include/searcher_amd.h (lcs::track): reads "rows n" then n x 9 doubles,
// then "codes m" and m ints, prints the three folds.  Driven by tests/test_tracker_pin.py against lte-cell-scanner_amd/tracker.py.
#include <cstdio>
#include <vector>
#include "../../include/searcher_amd.h"
int main() {
  int n = 0, m = 0;
  double f0 = 0, t0 = 0;
  if (std::scanf("%lf %lf %d", &f0, &t0, &n) != 3) return 1;
  std::vector<double> meas((size_t)n * LCS_TRK_MEAS);
  for (double &v : meas) if (std::scanf("%lf", &v) != 1) return 1;
  if (std::scanf("%d", &m) != 1) return 1;
  std::vector<int32_t> codes(m);
  for (int32_t &v : codes) if (std::scanf("%d", &v) != 1) return 1;
  const lcs::track::MibLock s = lcs::track::mib_lock_walk(codes.data(), m);
  std::printf("%.17g %.17g %.17g %d %d %d\n", lcs::track::fold_frequency_offset(f0, meas.data(), n), lcs::track::fold_frame_timing(t0, meas.data(), n),
              s.failures, (int)s.synchronized, s.attempts, (int)s.dropped);
  return 0;
}
