// test_shim.cpp -- drives the reference-shaped free functions of host/searcher_shim.cpp the way the reference's
// main loop does (src/CellSearch.cpp:484-558: correlate, threshold, peak search, then per peak SSS -> FOE -> grid ->
// TFOEC -> MIB) on one recorded capture buffer and prints what it decodes.  Used by tests/test_cli.py.
//   test_shim <capbuf_NNNN.it> [ppm]      one line per decoded cell
//   test_shim --del-oob                   host-only check of del_oob (no GPU needed)
//   test_shim --xc <capbuf.it> <out.bin> [f0 f1 ...]   xcorr_pss with lcs_shim_want_xc(true): every output flattened the way
//                                         test/test_xcorr_pss.cpp:104-124 flattens them, written as raw doubles / ints
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <stdexcept>
#include <stdint.h>

#include "../include/lcs.h"
#include "itfile.hpp"
#include "searcher_shim.h"

static int check_del_oob() {
  const int in[] = {-1, 0, 5, 12, 11, 13, -7, 3};
  itpp::ivec v(8);
  for (int i = 0; i < 8; ++i) v(i) = in[i];
  del_oob(v);
  const int want[] = {0, 5, 11, 3};
  bool ok = v.length() == 4;
  for (int i = 0; ok && i < 4; ++i) ok = v(i) == want[i];
  itpp::ivec e(0);
  del_oob(e);
  ok = ok && e.length() == 0;
  std::printf("del_oob %s\n", ok ? "ok" : "FAILED");
  return ok ? 0 : 1;
}

// test/test_xcorr_pss.cpp:94-124 through the shim: call xcorr_pss with the reference's argument list, flatten xc, sp,
// sp_incoherent, xc_incoherent_single, xc_incoherent, the collapsed powers and frequency indices exactly as the
// reference's test does (itpp_ext::flatten: first index fastest; cvectorize: column-major) and dump them; the Python
// side applies the reference test's tolerances against the oracle.
static void put(std::FILE *f, const void *p, size_t bytes) { if (std::fwrite(p, 1, bytes, f) != bytes) throw std::runtime_error("short write"); }
static int dump_xc(int argc, char **argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: test_shim --xc <capbuf.it> <out.bin> [f ...]\n"); return 2; }
  std::map<std::string, itfile::Var> vars = itfile::read_all(argv[2]);
  const std::vector<std::complex<double> > samples = itfile::get_dcvec(vars, "capbuf");
  const double fc = (double)itfile::get_ivec(vars, "fc").at(0);
  itpp::cvec capbuf((int)samples.size());
  for (size_t i = 0; i < samples.size(); ++i) capbuf((int)i) = samples[i];
  itpp::vec f_search_set(argc > 4 ? argc - 4 : 1);
  f_search_set(0) = 0.0;
  for (int i = 4; i < argc; ++i) f_search_set(i - 4) = std::atof(argv[i]);
  itpp::mat pow;
  itpp::imat frq;
  vf3d single, incoherent;
  itpp::vec sp_incoherent, sp;
  vcf3d xc;
  uint16 n_comb_xc = 0, n_comb_sp = 0;
  xcorr_pss(capbuf, f_search_set, 2, fc, fc, 1.92e6, pow, frq, single, incoherent, sp_incoherent, xc, sp, n_comb_xc, n_comb_sp);
  const bool empty_by_default = xc.empty();
  lcs_shim_want_xc(true);
  xcorr_pss(capbuf, f_search_set, 2, fc, fc, 1.92e6, pow, frq, single, incoherent, sp_incoherent, xc, sp, n_comb_xc, n_comb_sp);
  std::FILE *f = std::fopen(argv[3], "wb");
  if (!f) throw std::runtime_error("cannot write the output file");
  const int64_t hdr[8] = {(int64_t)xc.size(), (int64_t)xc.at(0).size(), (int64_t)xc.at(0).at(0).size(), sp.length(), sp_incoherent.length(),
                          n_comb_xc, n_comb_sp, empty_by_default};
  put(f, hdr, sizeof(hdr));
  for (size_t d3 = 0; d3 < xc[0][0].size(); ++d3)            // itpp_ext::flatten(vcf3d), src/itpp_ext.cpp:37-62
    for (size_t d2 = 0; d2 < xc[0].size(); ++d2)
      for (size_t d1 = 0; d1 < xc.size(); ++d1) { const std::complex<double> v(xc[d1][d2][d3]); put(f, &v, sizeof(v)); }
  put(f, sp._data(), sizeof(double) * sp.length());
  put(f, sp_incoherent._data(), sizeof(double) * sp_incoherent.length());
  const vf3d *both[2] = {&single, &incoherent};
  for (int w = 0; w < 2; ++w)
    for (size_t d3 = 0; d3 < (*both[w])[0][0].size(); ++d3)
      for (size_t d2 = 0; d2 < (*both[w])[0].size(); ++d2)
        for (size_t d1 = 0; d1 < both[w]->size(); ++d1) { const double v = (*both[w])[d1][d2][d3]; put(f, &v, sizeof(v)); }
  for (int c = 0; c < pow.cols(); ++c) for (int r = 0; r < pow.rows(); ++r) put(f, &pow(r, c), sizeof(double));      // cvectorize
  for (int c = 0; c < frq.cols(); ++c) for (int r = 0; r < frq.rows(); ++r) { const int64_t v = frq(r, c); put(f, &v, sizeof(v)); }
  std::fclose(f);
  lcs_shim_want_xc(false);
  std::printf("xc %zu x %zu x %zu n_comb_xc %d n_comb_sp %d\n", xc.size(), xc[0].size(), xc[0][0].size(), (int)n_comb_xc, (int)n_comb_sp);
  return 0;
}

int main(int argc, char **argv) {
  if (argc >= 2 && !std::strcmp(argv[1], "--del-oob")) return check_del_oob();
  if (argc >= 2 && !std::strcmp(argv[1], "--xc")) {
    try { return dump_xc(argc, argv); } catch (const std::exception &e) { std::fprintf(stderr, "Error: %s\n", e.what()); return 1; }
  }
  if (argc < 2) { std::fprintf(stderr, "usage: test_shim <capbuf.it> [ppm] | --del-oob\n"); return 2; }
  try {
    std::map<std::string, itfile::Var> vars = itfile::read_all(argv[1]);
    const std::vector<std::complex<double> > samples = itfile::get_dcvec(vars, "capbuf");
    const double fc = (double)itfile::get_ivec(vars, "fc").at(0);
    const double ppm = argc >= 3 ? std::atof(argv[2]) : 120.0;
    itpp::cvec capbuf((int)samples.size());
    for (size_t i = 0; i < samples.size(); ++i) capbuf((int)i) = samples[i];
    const int n_extra = (int)std::floor((fc * ppm / 1e6 + 2.5e3) / 5e3);
    itpp::vec f_search_set(2 * n_extra + 1);
    for (int i = 0; i <= 2 * n_extra; ++i) f_search_set(i) = 5000.0 * (i - n_extra);
    const double fs = 1.92e6;
    const uint8 ds_comb_arm = 2;

    itpp::mat pow;
    itpp::imat frq;
    vf3d single, incoherent;
    itpp::vec sp_incoherent, sp;
    vcf3d xc;
    uint16 n_comb_xc = 0, n_comb_sp = 0;
    xcorr_pss(capbuf, f_search_set, ds_comb_arm, fc, fc, fs, pow, frq, single, incoherent, sp_incoherent, xc, sp, n_comb_xc, n_comb_sp);

    // detection threshold of the main loop (src/CellSearch.cpp:500-503)
    const double R_th1 = lcs_chi2cdf_inv(1 - 1e-12, 2.0 * n_comb_xc * (2 * ds_comb_arm + 1));
    const double rx_cutoff = (6 * 12 * 15e3 / 2 + 4 * 15e3) / (30720000.0 / 16 / 2);
    itpp::vec Z_th1(sp_incoherent.length());
    for (int i = 0; i < Z_th1.length(); ++i) Z_th1(i) = R_th1 * sp_incoherent(i) / rx_cutoff / 137 / 2 / n_comb_xc / (2 * ds_comb_arm + 1);

    std::list<Cell> peaks;
    peak_search(pow, frq, Z_th1, f_search_set, fc, fc, single, ds_comb_arm, peaks);
    std::printf("n_comb_xc %d n_comb_sp %d peaks %d\n", (int)n_comb_xc, (int)n_comb_sp, (int)peaks.size());
    for (std::list<Cell>::iterator pk = peaks.begin(); pk != peaks.end(); ++pk) {
      itpp::vec h1np, h2np;
      itpp::cvec h1n, h2n, h1e, h2e;
      itpp::mat lln, lle;
      Cell c = sss_detect(*pk, capbuf, 3.0, fc, fc, fs, h1np, h2np, h1n, h2n, h1e, h2e, lln, lle);
      if (c.n_id_1 == -1) continue;                              // no SSS: the reference erases the peak (:530-534)
      c = pss_sss_foe(c, capbuf, fc, fc, fs);
      itpp::cmat tfg, tfg_comp;
      itpp::vec ts, ts_comp;
      extract_tfg(c, capbuf, fc, fc, fs, tfg, ts);
      const RS_DL rs_dl(c.n_id_cell(), 6, c.cp_type);
      c = tfoec(c, tfg, ts, fc, fc, rs_dl, tfg_comp, ts_comp);
      c = decode_mib(c, tfg_comp, rs_dl);
      if (c.n_rb_dl == -1) continue;                             // no MIB (:554-558)
      std::printf("cell %d ports %d n_rb_dl %d cp %d phich_dur %d phich_res %d sfn %d freq_superfine %.3f tfg_rows %d\n",
                  (int)c.n_id_cell(), (int)c.n_ports, (int)c.n_rb_dl, (int)c.cp_type, (int)c.phich_duration,
                  (int)c.phich_resource, (int)c.sfn, c.freq_superfine, tfg.rows());
    }
  } catch (const std::exception &e) {
    std::fprintf(stderr, "Error: %s\n", e.what());
    return 1;
  }
  return 0;
}
