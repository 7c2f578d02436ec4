// searcher_shim.cpp -- definitions of the reference's searcher functions (include/searcher.h:22-124) over
// liblcs_amd.so.  This is the file a maintainer compiles INSTEAD of src/searcher.cpp: every caller of the reference
// (src/CellSearch.cpp:497-553, src/LTE-Tracker.cpp:632-683, src/searcher_thread.cpp:120-196, test/*.cpp) links
// against these symbols unchanged.
//
// One lcs_ctx per calling thread, created on first use: the reference's functions are re-entrant and are called
// from the main thread and from the searcher thread; a context serialises its own calls.
#include "searcher_shim.h"

#include <atomic>
#include <memory>

#define LCS_CONTAINER_NS itpp
#include "../include/searcher_amd.h"

namespace {

std::atomic<int> g_device(-1);
std::atomic<bool> g_want_xc(false);

lcs::Searcher &gpu() {
  // lives as long as the thread and is destroyed (streams, events, workspace) when the thread exits; a failing
  // lcs_create throws lcs::error to the caller and is tried again by the next call
  static thread_local std::unique_ptr<lcs::Searcher> s;
  if (!s) s.reset(new lcs::Searcher(g_device.load()));
  return *s;
}

// reference Cell <-> the C ABI's POD record (include/lcs.h); enums travel as their integer values, which the two
// sides define identically (common.h.in:48-96 / LCS_CP_*)
lcs::Cell to_lcs(const Cell &c) {
  lcs::Cell o;
  o.fc_requested = c.fc_requested; o.fc_programmed = c.fc_programmed; o.pss_pow = c.pss_pow;
  o.ind = c.ind; o.freq = c.freq; o.n_id_2 = c.n_id_2; o.n_id_1 = c.n_id_1; o.cp_type = (int)c.cp_type;
  o.frame_start = c.frame_start; o.freq_fine = c.freq_fine; o.freq_superfine = c.freq_superfine;
  o.n_ports = c.n_ports; o.n_rb_dl = c.n_rb_dl; o.phich_duration = (int)c.phich_duration;
  o.phich_resource = (int)c.phich_resource; o.sfn = c.sfn;
  return o;
}

Cell from_lcs(const lcs_cell &c) {
  Cell o;
  o.fc_requested = c.fc_requested; o.fc_programmed = c.fc_programmed; o.pss_pow = c.pss_pow;
  o.ind = c.ind; o.freq = c.freq; o.n_id_2 = (int8)c.n_id_2; o.n_id_1 = (int16)c.n_id_1;
  o.cp_type = (cp_type_t::cp_type_t)c.cp_type;
  o.frame_start = c.frame_start; o.freq_fine = c.freq_fine; o.freq_superfine = c.freq_superfine;
  o.n_ports = (int8)c.n_ports; o.n_rb_dl = (int8)c.n_rb_dl;
  o.phich_duration = (phich_duration_t::phich_duration_t)c.phich_duration;
  o.phich_resource = (phich_resource_t::phich_resource_t)c.phich_resource;
  o.sfn = (int16)c.sfn;
  return o;
}

}  // namespace

// Applies to contexts created AFTER the call: set it before a thread's first searcher call (a thread that already has
// its context keeps its device).
void lcs_shim_set_device(int device) { g_device.store(device); }

// The raw correlations `xc` (include/searcher.h:35) are a debug output only test/test_xcorr_pss.cpp:104-109 reads --
// 136 MB at n_f = 37, computed by a separate fp64 kernel with the reference's arithmetic (complex<double> accumulation,
// complex<float> storage, src/searcher.cpp:136,167-169).  Off by default (`xc` comes back empty, as no other caller
// looks at it); a caller that wants them switches them on here.  Process-wide, applies from the next xcorr_pss call.
void lcs_shim_want_xc(bool on) { g_want_xc.store(on); }

void xcorr_pss(const itpp::cvec &capbuf, const itpp::vec &f_search_set, const uint8 &ds_comb_arm,
               const double &fc_requested, const double &fc_programmed, const double &fs_programmed,
               itpp::mat &xc_incoherent_collapsed_pow, itpp::imat &xc_incoherent_collapsed_frq,
               vf3d &xc_incoherent_single, vf3d &xc_incoherent, itpp::vec &sp_incoherent, vcf3d &xc, itpp::vec &sp,
               uint16 &n_comb_xc, uint16 &n_comb_sp) {
  unsigned short ncx = 0, ncs = 0;
  // `xc` is filled only after lcs_shim_want_xc(true) (see there); otherwise it comes back empty
  gpu().xcorr_pss(capbuf, f_search_set, ds_comb_arm, fc_requested, fc_programmed, fs_programmed,
                  xc_incoherent_collapsed_pow, xc_incoherent_collapsed_frq, xc_incoherent_single, xc_incoherent,
                  sp_incoherent, xc, sp, ncx, ncs, g_want_xc.load());
  n_comb_xc = ncx;
  n_comb_sp = ncs;
}

void peak_search(const itpp::mat &xc_incoherent_collapsed_pow, const itpp::imat &xc_incoherent_collapsed_frq,
                 const itpp::vec &Z_th1, const itpp::vec &f_search_set, const double &fc_requested,
                 const double &fc_programmed, const vf3d &xc_incoherent_single, const uint8 &ds_comb_arm,
                 std::list<Cell> &cells) {
  std::list<lcs::Cell> found;
  gpu().peak_search(xc_incoherent_collapsed_pow, xc_incoherent_collapsed_frq, Z_th1, f_search_set, fc_requested,
                    fc_programmed, xc_incoherent_single, ds_comb_arm, found);
  for (std::list<lcs::Cell>::const_iterator it = found.begin(); it != found.end(); ++it) cells.push_back(from_lcs(*it));
}

Cell sss_detect(const Cell &cell, const itpp::cvec &capbuf, const double &thresh2_n_sigma, const double &fc_requested,
                const double &fc_programmed, const double &fs_programmed, itpp::vec &sss_h1_np_est,
                itpp::vec &sss_h2_np_est, itpp::cvec &sss_h1_nrm_est, itpp::cvec &sss_h2_nrm_est,
                itpp::cvec &sss_h1_ext_est, itpp::cvec &sss_h2_ext_est, itpp::mat &log_lik_nrm, itpp::mat &log_lik_ext) {
  return from_lcs(gpu().sss_detect(to_lcs(cell), capbuf, thresh2_n_sigma, fc_requested, fc_programmed, fs_programmed,
                                   sss_h1_np_est, sss_h2_np_est, sss_h1_nrm_est, sss_h2_nrm_est, sss_h1_ext_est,
                                   sss_h2_ext_est, log_lik_nrm, log_lik_ext));
}

Cell pss_sss_foe(const Cell &cell_in, const itpp::cvec &capbuf, const double &fc_requested, const double &fc_programmed,
                 const double &fs_programmed) {
  return from_lcs(gpu().pss_sss_foe(to_lcs(cell_in), capbuf, fc_requested, fc_programmed, fs_programmed));
}

void extract_tfg(const Cell &cell, const itpp::cvec &capbuf_raw, const double &fc_requested, const double &fc_programmed,
                 const double &fs_programmed, itpp::cmat &tfg, itpp::vec &tfg_timestamp) {
  gpu().extract_tfg(to_lcs(cell), capbuf_raw, fc_requested, fc_programmed, fs_programmed, tfg, tfg_timestamp);
}

// rs_dl is a pure function of (n_id_cell, cp_type) (src/lte_lib.cpp:305-405, built right before these calls at
// src/CellSearch.cpp:545): the device rebuilds it, the argument is accepted and ignored
Cell tfoec(const Cell &cell, const itpp::cmat &tfg, const itpp::vec &tfg_timestamp, const double &fc_requested,
           const double &fc_programmed, const RS_DL &rs_dl, itpp::cmat &tfg_comp, itpp::vec &tfg_comp_timestamp) {
  (void)rs_dl;
  return from_lcs(gpu().tfoec(to_lcs(cell), tfg, tfg_timestamp, fc_requested, fc_programmed, tfg_comp, tfg_comp_timestamp));
}

Cell decode_mib(const Cell &cell, const itpp::cmat &tfg, const RS_DL &rs_dl) {
  (void)rs_dl;
  return from_lcs(gpu().decode_mib(to_lcs(cell), tfg));
}

// src/searcher.cpp:1072-1083: drop the entries outside 0..11 (LTE-Tracker's OFDM symbol bookkeeping)
void del_oob(itpp::ivec &v) {
  int kept = 0;
  for (int t = 0; t < v.length(); ++t)
    if (v(t) >= 0 && v(t) <= 11) v(kept++) = v(t);
  v.set_size(kept, true);
}
