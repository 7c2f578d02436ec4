// StreamSearch -- the consumer loop of LTE-Tracker's searcher thread (ref src/searcher_thread.cpp:83-246) on top of the
// streaming entry points of liblcs_amd.so, in C++.
//
// The reference's thread waits for an 80 ms capture buffer, searches it with ONE frequency hypothesis (the tracker's
// current offset, :97-98), skips every peak whose cell is already tracked (:157-177), decodes the others and hands new
// cells to the tracker (:200-245); then it takes the next buffer.  Here the buffers come from capbuf_NNNN.it files (the
// reference's hardware-free input), the chain is the hipGraph lcs_stream_open captured, and two buffers are kept in flight:
// buffer i + 1 is copied to pinned memory and launched while buffer i is still on the GPU.
//
//   StreamSearch [-g gpu] [-f f_off_hz] [-n passes] <capbuf_0000.it> [more .it files ...]
// prints one line per buffer: new cells (n_id_cell, frame timing on the tracker's 1.92 MHz time base, frequency offset)
// and the number of already tracked cells seen again; the tracked list grows as cells are found, as in the reference.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <string>
#include <vector>

#include "../include/searcher_amd.h"
#include "itfile.hpp"

int main(int argc, char **argv) {
  int gpu = -1, passes = 1;
  double f_off = 0.0;
  std::vector<std::string> files;
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "-g") && i + 1 < argc) gpu = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "-f") && i + 1 < argc) f_off = std::atof(argv[++i]);
    else if (!std::strcmp(argv[i], "-n") && i + 1 < argc) passes = std::atoi(argv[++i]);
    else files.push_back(argv[i]);
  }
  if (files.empty()) { std::fprintf(stderr, "usage: StreamSearch [-g gpu] [-f f_off_hz] [-n passes] capbuf_0000.it [...]\n"); return 2; }
  try {
    // recorded captures are (u8 - 127) / 128 exactly: they travel as bytes (8x less PCIe than complex<double>)
    std::vector<std::vector<unsigned char> > bufs;
    double fc = 0;
    size_t n_cap = 0;
    for (size_t k = 0; k < files.size(); ++k) {
      std::map<std::string, itfile::Var> vars = itfile::read_all(files[k]);
      const std::vector<std::complex<double> > s = itfile::get_dcvec(vars, "capbuf");
      const std::vector<int32_t> fcv = itfile::get_ivec(vars, "fc");
      if (k == 0) { n_cap = s.size(); fc = fcv.empty() ? 0.0 : (double)fcv[0]; }
      if (s.size() != n_cap) { std::fprintf(stderr, "Error: %s has a different length\n", files[k].c_str()); return 1; }
      std::vector<unsigned char> b(2 * n_cap);
      const double *x = reinterpret_cast<const double *>(s.data());
      for (size_t i = 0; i < b.size(); ++i) {
        const double code = x[i] * 128.0 + 127.0;
        if (!(code >= 0.0 && code <= 255.0) || code != std::floor(code)) { std::fprintf(stderr, "Error: %s is not a dongle capture\n", files[k].c_str()); return 1; }
        b[i] = (unsigned char)code;
      }
      bufs.push_back(b);
    }
    const double fs = 1.92e6;
    lcs::Searcher searcher(gpu);
    searcher.stream_open(LCS_FMT_IQ_U8, (uint32_t)n_cap, fc, fc, fs);
    std::vector<int16_t> tracked;
    const int total = passes * (int)bufs.size();
    std::vector<size_t> n_tracked_at_push(total);
    float gpu_ms_sum = 0;
    auto collect = [&](int i) {
      std::list<lcs::Cell> fresh;
      float ms = 0;
      const int again = searcher.stream_collect(fresh, &ms);
      gpu_ms_sum += ms;
      std::printf("buffer %d: %d new, %d tracked cell(s) seen again", i, (int)fresh.size(), again);
      for (std::list<lcs::Cell>::const_iterator c = fresh.begin(); c != fresh.end(); ++c) {
        // the tracker's time base: frame_start * (FS_LTE/16) / (fs_programmed * k_factor)  (ref :224, capture latency 0)
        const double k_factor = (fc - c->freq_superfine) / fc;
        std::printf("  [cell %d ports %d nRB %d frame_timing %.3f f_off %.2f]", c->n_id_cell(), c->n_ports, c->n_rb_dl,
                    c->frame_start * (30.72e6 / 16) / (fs * k_factor), c->freq_superfine);
        bool known = false;
        for (size_t t = 0; t < tracked.size(); ++t) known = known || tracked[t] == c->n_id_cell();
        if (!known) tracked.push_back((int16_t)c->n_id_cell());        // ref :200-245: a new tracker thread is started
      }
      std::printf("\n");
    };
    for (int i = 0; i < total; ++i) {
      n_tracked_at_push[i] = tracked.size();
      searcher.stream_push(bufs[i % bufs.size()].data(), f_off, tracked);
      if (i >= 1) collect(i - 1);
    }
    collect(total - 1);
    searcher.stream_close();
    std::printf("tracked:");
    for (size_t t = 0; t < tracked.size(); ++t) std::printf(" %d", tracked[t]);
    std::printf("\n%d buffers, %.3f ms of GPU time per buffer\n", total, gpu_ms_sum / total);
  } catch (const std::exception &e) {
    std::fprintf(stderr, "Error: %s\n", e.what());
    return 2;
  }
  return 0;
}
