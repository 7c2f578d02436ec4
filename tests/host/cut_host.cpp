// Host-side harness for the producer thread's symbol cutter of lte_device.h (the same __host__ __device__ code k_trk_cut_hits /
// k_trk_cut_walk run): tests/test_track_cut_host.py compares the closed form with the sample-by-sample walk and with
// lte-cell-scanner_amd/tracker.py's cutter.  Test infrastructure.
#include "../../lte-cell-scanner_amd/csrc/lte_device.h"

// closed form for every symbol; returns 1 when the premise held for all of them (then hit / late are final), 0 otherwise
extern "C" int cut_host_closed(int cp_type, double frame_timing, double freq_off, double fc_req, double fc_prog, double fs_prog,
                               unsigned n_cap, int n_sym, double ts0, long long k0, long long pos0, int *hit, double *late) {
  const TrkCutCell q = trk_cut_cell(cp_type, frame_timing, freq_off, fc_req, fc_prog, fs_prog, ts0, (long)k0, (long)pos0);
  double l0;
  const long h0 = trk_cut_first(q, n_cap, &l0);
  int ok = 1;
  for (int k = 0; k < n_sym; ++k) {
    long h;
    double lt;
    if (!trk_cut_symbol(q, n_cap, k, h0, l0, &h, &lt)) ok = 0;
    hit[k] = (int)h;
    late[k] = lt;
  }
  // what k_trk_cut_walk does behind an unflagged cell: nothing behind the first symbol that does not fit
  int n = 0;
  while (n < n_sym && hit[n] >= 0) ++n;
  for (int k = n; k < n_sym; ++k) { hit[k] = -1; late[k] = 0.0; }
  return ok;
}
extern "C" int cut_host_walk(int cp_type, double frame_timing, double freq_off, double fc_req, double fc_prog, double fs_prog,
                             unsigned n_cap, int n_sym, double ts0, long long k0, long long pos0, int *hit, double *late) {
  return trk_cut_walk(trk_cut_cell(cp_type, frame_timing, freq_off, fc_req, fc_prog, fs_prog, ts0, (long)k0, (long)pos0), n_cap, n_sym, hit, late);
}
