// TrackCells -- the producer thread and the tracker threads' loop of LTE-Tracker on a recorded capture, in C++ on top of
// liblcs_amd.so: the cells of the buffer are found with the searcher (as LTE-Tracker's main thread does before it starts
// tracking, ref src/LTE-Tracker.cpp:632-683), then every cell's OFDM symbols are cut out of the sample stream the way the
// producer thread does (ref src/producer_thread.cpp:96-131, 196-246: the first sample whose timestamp on the cell-independent
// 1.92 MHz time base is within half a sample of the symbol's target) and handed to the tracker block after block
// (lcs_track_stream_block = get_fd, reference-symbol channel estimates, do_foe / do_toe_v2 measurements, MIB re-decode:
// ref src/tracker_thread.cpp:823-1068).  The three slow loops those measurements feed -- global frequency offset, frame
// timing, MIB lock -- are the scalar recurrences of lcs::track (include/searcher_amd.h).
//
//   TrackCells [-g gpu] [-b symbols_per_block] [-p ppm] [-D] <capbuf_0000.it>
// prints one line per (block, cell): symbols consumed, frequency offset, frame timing, MIB lock state.
// -D: the symbols are cut ON THE DEVICE (lcs_track_cut: the capture is uploaded once, the symbols of every cell stay in HBM and
// lcs_track_stream_block reads them there); the whole stream then goes through as ONE block.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <string>
#include <vector>

#include "../include/searcher_amd.h"
#include "itfile.hpp"

namespace {

const double FS_LTE = 30720000.0;

// the symbols of one tracked cell, as queued by the producer
struct CellFeed {
  lcs::Cell cell;
  std::vector<std::complex<double> > td;   // [n][128]
  std::vector<double> late;                // [n]
  double frame_timing, frequency_offset;   // the values in force while the symbols were cut (open loop on a recorded buffer)
  int n;
};

// ref src/producer_thread.cpp:96-131 (timestamps), :196-246 (symbol extraction); the buffer's first sample has timestamp 0
void cut_symbols(const std::vector<std::complex<double> > &cap, double fc_requested, double fc_programmed, double fs_programmed,
                 CellFeed &f) {
  const double k_factor = (fc_requested - f.frequency_offset) / fc_programmed;
  const double step = (FS_LTE / 16) / (fs_programmed * k_factor);
  const bool normal = f.cell.cp_type == LCS_CP_NORMAL;
  const int nsd = normal ? 7 : 6;
  double target = normal ? 10.0 : 32.0;
  int sym = 0;
  size_t pos = 0;
  f.n = 0;
  while (pos + 128 <= cap.size()) {
    const size_t limit = std::min(cap.size() - 128, pos + 25000);
    long hit = -1;
    double hit_late = 0;
    for (size_t n = pos; n <= limit; ++n) {
      const double ts = lcs::track::wrap((double)n * step, 0.0, 19200.0);
      const double tdiff = lcs::track::wrap(ts - (f.frame_timing + target), -9600.0, 9600.0);
      if (std::fabs(tdiff) < 0.5 || (tdiff > 0 && tdiff < 3)) { hit = (long)n; hit_late = tdiff; break; }
    }
    if (hit < 0) break;
    f.td.insert(f.td.end(), cap.begin() + hit, cap.begin() + hit + 128);
    f.late.push_back(hit_late);
    ++f.n;
    pos = (size_t)hit + 128;
    target = std::fmod(target + (normal ? (sym == 6 ? 138.0 : 137.0) : 160.0), 19200.0);
    sym = (sym + 1) % nsd;
  }
}

}  // namespace

int main(int argc, char **argv) {
  int gpu = -1, block = 140;
  double ppm = 120.0;
  bool device_cut = false;
  std::string file;
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "-g") && i + 1 < argc) gpu = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "-b") && i + 1 < argc) block = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "-p") && i + 1 < argc) ppm = std::atof(argv[++i]);
    else if (!std::strcmp(argv[i], "-D")) device_cut = true;
    else file = argv[i];
  }
  if (file.empty() || block < 1) { std::fprintf(stderr, "usage: TrackCells [-g gpu] [-b symbols_per_block] [-p ppm] [-D] capbuf_0000.it\n"); return 2; }
  try {
    std::map<std::string, itfile::Var> vars = itfile::read_all(file);
    const std::vector<std::complex<double> > cap = itfile::get_dcvec(vars, "capbuf");
    const std::vector<int32_t> fcv = itfile::get_ivec(vars, "fc");
    const double fc = fcv.empty() ? 0.0 : (double)fcv[0], fs = 1.92e6;

    // the searcher's frequency grid for this crystal tolerance (ref src/CellSearch.cpp:463-465)
    const int n_extra = (int)std::floor((fc * ppm / 1e6 + 2.5e3) / 5e3);
    lcs::cn::vec f_search_set(2 * n_extra + 1);
    for (int k = -n_extra; k <= n_extra; ++k) f_search_set(k + n_extra) = 5e3 * k;
    lcs::cn::cvec capv((int)cap.size());
    for (size_t i = 0; i < cap.size(); ++i) capv((int)i) = cap[i];

    lcs::Searcher searcher(gpu);
    std::list<lcs::Cell> found;
    searcher.search_capbuf(capv, f_search_set, fc, fc, fs, found);
    std::vector<CellFeed> feeds;
    for (std::list<lcs::Cell>::const_iterator c = found.begin(); c != found.end(); ++c) {
      if (c->n_rb_dl <= 0) continue;                         // only cells whose MIB was decoded are tracked (ref LTE-Tracker.cpp:683)
      CellFeed f;
      f.cell = *c;
      f.frequency_offset = c->freq_superfine;
      const double k_factor = (fc - c->freq_superfine) / fc;
      f.frame_timing = c->frame_start * (FS_LTE / 16) / (fs * k_factor);       // ref src/searcher_thread.cpp:224
      cut_symbols(cap, fc, fc, fs, f);
      feeds.push_back(f);
    }
    if (feeds.empty()) { std::printf("no cell to track\n"); return 0; }
    int n_total = feeds[0].n;
    for (size_t i = 1; i < feeds.size(); ++i) n_total = std::min(n_total, feeds[i].n);
    const int C = (int)feeds.size();
    // -D: the same symbols cut on the device.  The capture goes up once (complex<double>, as the file holds it); a first call finds
    // how many symbols every cell has in the buffer, the second leaves [cell][n_total][128] in HBM for ONE block over the whole stream.
    void *d_cap = 0, *d_td = 0;
    std::vector<double> dev_late;
    if (device_cut) {
      std::vector<int32_t> cp(C), n_cut;
      std::vector<double> ft(C), fo(C);
      for (int i = 0; i < C; ++i) { cp[i] = feeds[i].cell.cp_type; ft[i] = feeds[i].frame_timing; fo[i] = feeds[i].frequency_offset; }
      const size_t cap_bytes = cap.size() * sizeof(std::complex<double>);
      const int probe = (int)(cap.size() / 137 + 2);
      if (lcs_device_alloc(searcher.handle(), cap_bytes, &d_cap) != LCS_OK || lcs_device_upload(searcher.handle(), d_cap, cap.data(), cap_bytes) != LCS_OK ||
          lcs_device_alloc(searcher.handle(), (size_t)C * probe * 128 * sizeof(std::complex<double>), &d_td) != LCS_OK)
        throw lcs::error(std::string("liblcs_amd: ") + lcs_last_error(searcher.handle()));
      searcher.track_cut(d_cap, LCS_FMT_C128, (uint32_t)cap.size(), cp, ft, fo, fc, fc, fs, probe, d_td, dev_late, n_cut);
      for (int i = 0; i < C; ++i)
        if (n_cut[i] != feeds[i].n) { std::fprintf(stderr, "Error: device cutter found %d symbols of cell %d, host cutter %d\n", n_cut[i], i, feeds[i].n); return 3; }
      searcher.track_cut(d_cap, LCS_FMT_C128, (uint32_t)cap.size(), cp, ft, fo, fc, fc, fs, n_total, d_td, dev_late, n_cut);
      for (int i = 0; i < C; ++i)
        for (int k = 0; k < n_total; ++k)
          if (dev_late[(size_t)i * n_total + k] != feeds[i].late[k]) { std::fprintf(stderr, "Error: device cutter: late differs at symbol %d of cell %d\n", k, i); return 3; }
      block = n_total;
      std::printf("device cutter: %d symbols of %d cell(s) in HBM\n", n_total, C);
    }
    std::printf("tracking %d cell(s), %d OFDM symbols each, %d per block\n", C, n_total, block);

    std::vector<lcs_track_cell> tc(C);
    std::vector<double> f_off(C), f_tim(C);
    std::vector<lcs::track::MibLock> lock(C);
    std::vector<std::vector<int32_t> > codes(C);             // mib_ok of every frame offset seen so far
    for (int i = 0; i < C; ++i) {
      const lcs::Cell &c = feeds[i].cell;
      std::memset(&tc[i], 0, sizeof(tc[i]));
      tc[i].n_id_1 = c.n_id_1; tc[i].n_id_2 = c.n_id_2; tc[i].cp_type = c.cp_type; tc[i].n_ports = c.n_ports; tc[i].n_rb_dl = c.n_rb_dl;
      tc[i].phich_duration = c.phich_duration; tc[i].phich_resource = c.phich_resource;
      f_off[i] = feeds[i].frequency_offset;
      f_tim[i] = feeds[i].frame_timing;
      lock[i].failures = 0; lock[i].synchronized = false; lock[i].attempts = 0; lock[i].dropped = false;
    }
    lcs::Searcher::TrackRows rows;
    for (int s0 = 0; s0 < n_total; s0 += block) {
      const int n = std::min(block, n_total - s0);
      std::vector<std::complex<double> > td((size_t)C * n * 128);
      std::vector<double> fo((size_t)C * n), ft((size_t)C * n), lt((size_t)C * n);
      for (int i = 0; i < C; ++i) {
        std::copy(feeds[i].td.begin() + (size_t)s0 * 128, feeds[i].td.begin() + (size_t)(s0 + n) * 128, td.begin() + (size_t)i * n * 128);
        for (int k = 0; k < n; ++k) { fo[(size_t)i * n + k] = feeds[i].frequency_offset; ft[(size_t)i * n + k] = feeds[i].frame_timing; lt[(size_t)i * n + k] = feeds[i].late[s0 + k]; }
      }
      searcher.track_stream_block(tc, n, device_cut ? static_cast<const std::complex<double> *>(d_td) : td.data(), fo.data(), ft.data(), lt.data(), fc, fc, fs, rows);
      for (int i = 0; i < C; ++i) {
        // port 0's filtered reference symbols drive the two loops (every port measures the same offsets; the reference
        // runs the recurrences in symbol order over the ports it tracks, port 0 first)
        f_off[i] = lcs::track::fold_frequency_offset(f_off[i], rows.meas_rows(i, 0), rows.n_meas[(size_t)i * 4]);
        f_tim[i] = lcs::track::fold_frame_timing(f_tim[i], rows.meas_rows(i, 0), rows.n_meas[(size_t)i * 4]);
        for (int k = 0; k < rows.n_mib[i]; ++k) codes[i].push_back(rows.mib_ok[(size_t)i * rows.max_off + k]);
        lock[i] = lcs::track::mib_lock_walk(codes[i].data(), (int)codes[i].size());
        std::printf("symbols %5d  cell %3d  f_off %.6f  frame_timing %.6f  mib attempts %d failures %.2f %s\n", s0 + n,
                    feeds[i].cell.n_id_cell(), f_off[i], f_tim[i], lock[i].attempts, lock[i].failures,
                    lock[i].synchronized ? "LOCKED" : "searching");
      }
    }
    if (d_td) (void)lcs_device_free(searcher.handle(), d_td);
    if (d_cap) (void)lcs_device_free(searcher.handle(), d_cap);
  } catch (const std::exception &e) {
    std::fprintf(stderr, "Error: %s\n", e.what());
    return 2;
  }
  return 0;
}
