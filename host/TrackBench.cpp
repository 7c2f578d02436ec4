// TrackBench -- the tracker block pipeline (lcs_track_block: get_fd, reference-symbol channel estimates, do_foe / do_toe_v2
// measurements, MIB re-decode; ref src/tracker_thread.cpp:823-1068) driven the way LTE-Tracker drives its tracker threads:
// one host THREAD per context, each calling lcs_track_block in a loop on a block of tracked cells whose time-domain symbols
// are resident in HBM.  bench.py --stage track hands the block over in a file and reports this program's rate: the same
// library calls as its Python loop, without the interpreter between them (round 4's figure followed the box's host: 91-150 M
// symbols/s for the same 0.50 ms of GPU time per block).
//
//   TrackBench <block file> <contexts> <blocks> <warm-up blocks> [gpu] [capture bytes]
// With a capture file (the dongle's raw u8 I/Q of the buffer the block's symbols were cut from) every block STARTS FROM THE BYTES:
// they are uploaded once, each block then runs lcs_track_cut (the producer thread's symbol extraction on the device, every
// context into a symbol buffer of its own) followed by lcs_track_block on what it left in HBM; the cutter's `late` of the first
// block is compared with the block file's (the host cutter's), bit for bit.
// block file (little endian): int32 n_cells, n_sym; double fc_requested, fc_programmed, fs_programmed;
//   n_cells x lcs_track_cell; freq_off, frame_timing, late [n_cells][n_sym] doubles; td [n_cells][n_sym][128] complex<double>
// prints ONE JSON line.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../include/lcs.h"

namespace {
struct Block {
  int n_cells = 0, n_sym = 0;
  double fc_req = 0, fc_prog = 0, fs_prog = 0;
  std::vector<lcs_track_cell> cells;
  std::vector<double> fo, ft, late, td;
};

bool read_block(const char *path, Block &b) {
  FILE *f = std::fopen(path, "rb");
  if (!f) return false;
  int32_t hdr[2];
  double par[3];
  bool ok = std::fread(hdr, sizeof(hdr), 1, f) == 1 && std::fread(par, sizeof(par), 1, f) == 1;
  if (ok) {
    b.n_cells = hdr[0]; b.n_sym = hdr[1]; b.fc_req = par[0]; b.fc_prog = par[1]; b.fs_prog = par[2];
    const size_t N = (size_t)b.n_cells * b.n_sym;
    b.cells.resize(b.n_cells); b.fo.resize(N); b.ft.resize(N); b.late.resize(N); b.td.resize(N * 256);
    ok = std::fread(b.cells.data(), sizeof(lcs_track_cell), b.n_cells, f) == (size_t)b.n_cells && std::fread(b.fo.data(), 8, N, f) == N &&
         std::fread(b.ft.data(), 8, N, f) == N && std::fread(b.late.data(), 8, N, f) == N && std::fread(b.td.data(), 8, N * 256, f) == N * 256;
  }
  std::fclose(f);
  return ok;
}
}  // namespace

int main(int argc, char **argv) {
  if (argc < 5) { std::fprintf(stderr, "usage: TrackBench <block file> <contexts> <blocks> <warm-up blocks> [gpu] [capture bytes]\n"); return 2; }
  Block b;
  if (!read_block(argv[1], b)) { std::fprintf(stderr, "TrackBench: cannot read %s\n", argv[1]); return 2; }
  const int n_ctx = std::max(1, std::atoi(argv[2])), n_blocks = std::max(1, std::atoi(argv[3])), n_warm = std::max(0, std::atoi(argv[4]));
  const int gpu = argc > 5 ? std::atoi(argv[5]) : 0;
  const int max_rs = b.n_sym / 3 + 4, max_off = std::max(1, b.n_sym / 120 - 3);
  std::vector<lcs_ctx *> ctx(n_ctx, nullptr);
  for (int k = 0; k < n_ctx; ++k)
    if (lcs_create(gpu, &ctx[k]) != LCS_OK) { std::fprintf(stderr, "TrackBench: lcs_create failed (an MI355X is required)\n"); return 1; }
  void *d_td = nullptr;
  const size_t td_bytes = b.td.size() * sizeof(double);
  if (lcs_device_alloc(ctx[0], td_bytes, &d_td) != LCS_OK || lcs_device_upload(ctx[0], d_td, b.td.data(), td_bytes) != LCS_OK) {
    std::fprintf(stderr, "TrackBench: %s\n", lcs_last_error(ctx[0]));
    return 1;
  }
  // from the bytes: the capture in HBM, per context a symbol buffer the cutter fills, the per-cell values the symbols were cut with
  std::vector<unsigned char> cap_bytes;
  void *d_cap = nullptr;
  std::vector<void *> d_cut(n_ctx, nullptr);
  std::vector<int32_t> cp(b.n_cells);
  std::vector<double> ft0(b.n_cells), fo0(b.n_cells);
  std::atomic<bool> late_identical(true);
  if (argc > 6) {
    FILE *f = std::fopen(argv[6], "rb");
    if (!f) { std::fprintf(stderr, "TrackBench: cannot read %s\n", argv[6]); return 2; }
    std::fseek(f, 0, SEEK_END);
    cap_bytes.resize((size_t)std::ftell(f));
    std::fseek(f, 0, SEEK_SET);
    const bool ok = std::fread(cap_bytes.data(), 1, cap_bytes.size(), f) == cap_bytes.size();
    std::fclose(f);
    if (!ok || cap_bytes.size() < 256) { std::fprintf(stderr, "TrackBench: short capture file\n"); return 2; }
    if (lcs_device_alloc(ctx[0], cap_bytes.size(), &d_cap) != LCS_OK || lcs_device_upload(ctx[0], d_cap, cap_bytes.data(), cap_bytes.size()) != LCS_OK) {
      std::fprintf(stderr, "TrackBench: %s\n", lcs_last_error(ctx[0]));
      return 1;
    }
    for (int k = 0; k < n_ctx; ++k)
      if (lcs_device_alloc(ctx[k], td_bytes, &d_cut[k]) != LCS_OK) { std::fprintf(stderr, "TrackBench: %s\n", lcs_last_error(ctx[k])); return 1; }
    for (int i = 0; i < b.n_cells; ++i) { cp[i] = b.cells[i].cp_type; ft0[i] = b.ft[(size_t)i * b.n_sym]; fo0[i] = b.fo[(size_t)i * b.n_sym]; }
  }
  struct PerCtx {
    std::vector<double> late;
    std::vector<int32_t> n_cut;
    std::vector<lcs_track_cell> cells;
    std::vector<double> meas;
    std::vector<int32_t> n_meas, ce_upto, mib_ok;
    std::vector<uint64_t> mib_bits;
    double gpu_ms_sum = 0;
    int blocks = 0, locks = 0, failed = 0;
  };
  std::vector<PerCtx> pc(n_ctx);
  for (auto &p : pc) {
    p.cells = b.cells;
    p.meas.resize((size_t)b.n_cells * 4 * max_rs * LCS_TRK_MEAS);
    p.n_meas.resize((size_t)b.n_cells * 4); p.ce_upto.resize((size_t)b.n_cells * 4);
    p.mib_ok.resize((size_t)b.n_cells * max_off); p.mib_bits.resize((size_t)b.n_cells * max_off);
  }
  auto one_block = [&](int k) {
    PerCtx &p = pc[k];
    p.cells = b.cells;                                   // every block is a full, independent pass over the same symbols
    float ms = 0;
    const void *td = d_td;
    const double *late = b.late.data();
    if (d_cap) {
      p.late.resize((size_t)b.n_cells * b.n_sym);
      p.n_cut.resize(b.n_cells);
      if (lcs_track_cut(ctx[k], d_cap, LCS_FMT_IQ_U8, (uint32_t)(cap_bytes.size() / 2), 0.0, b.n_cells, cp.data(), ft0.data(), fo0.data(), nullptr, nullptr,
                        b.fc_req, b.fc_prog, b.fs_prog, b.n_sym, d_cut[k], p.late.data(), p.n_cut.data(), nullptr) != LCS_OK) { ++p.failed; return; }
      for (int i = 0; i < b.n_cells; ++i) if (p.n_cut[i] != b.n_sym) { ++p.failed; return; }
      if (p.blocks == 0 && std::memcmp(p.late.data(), b.late.data(), sizeof(double) * p.late.size()) != 0) late_identical = false;
      td = d_cut[k];
      late = p.late.data();
    }
    const int rc = lcs_track_block(ctx[k], p.cells.data(), b.n_cells, b.n_sym, td, 1, b.fo.data(), b.ft.data(), late, b.fc_req, b.fc_prog,
                                   b.fs_prog, nullptr, nullptr, nullptr, p.ce_upto.data(), p.meas.data(), max_rs, p.n_meas.data(), p.mib_ok.data(),
                                   p.mib_bits.data(), max_off, &ms);
    if (rc != LCS_OK) { ++p.failed; return; }
    p.gpu_ms_sum += ms;
    ++p.blocks;
    int l = 0;
    for (int32_t v : p.mib_ok) l += (v == 3);
    p.locks = l;
  };
  auto run = [&](int n) {
    std::atomic<int> next(0);
    std::vector<std::thread> th;
    for (int k = 0; k < n_ctx; ++k)
      th.emplace_back([&, k]() { while (next.fetch_add(1) < n) one_block(k); });
    for (auto &t : th) t.join();
  };
  run(std::max(n_warm, n_ctx));
  for (auto &p : pc) { p.gpu_ms_sum = 0; p.blocks = 0; }
  const auto t0 = std::chrono::steady_clock::now();
  run(n_blocks);
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  double gsum = 0;
  int done = 0, failed = 0;
  for (auto &p : pc) { gsum += p.gpu_ms_sum; done += p.blocks; failed += p.failed; }
  const double pipelined_ms = done ? gsum / done : 0;
  pc[0].gpu_ms_sum = 0; pc[0].blocks = 0;
  for (int i = 0; i < 5; ++i) one_block(0);               // the same block with nothing else on the GPU
  const double alone_ms = pc[0].blocks ? pc[0].gpu_ms_sum / pc[0].blocks : 0;
  std::printf("{\"symbols_per_s\": %.6g, \"blocks\": %d, \"failed\": %d, \"seconds\": %.6g, \"ms_per_block_wall\": %.6g, \"gpu_ms_per_block_pipelined\": %.6g, "
              "\"gpu_ms_per_block_alone\": %.6g, \"contexts\": %d, \"mib_locks_per_block\": %d, \"n_cells\": %d, \"n_sym\": %d, "
              "\"from_bytes\": %s, \"late_identical_to_host_cut\": %s}\n",
              (double)done * b.n_cells * b.n_sym / dt, done, failed, dt, 1e3 * dt / std::max(1, done), pipelined_ms, alone_ms, n_ctx, pc[0].locks,
              b.n_cells, b.n_sym, d_cap ? "true" : "false", (d_cap && late_identical.load()) ? "true" : "false");
  (void)lcs_device_free(ctx[0], d_td);
  if (d_cap) (void)lcs_device_free(ctx[0], d_cap);
  for (int k = 0; k < n_ctx; ++k) if (d_cut[k]) (void)lcs_device_free(ctx[k], d_cut[k]);
  for (lcs_ctx *c : ctx) lcs_destroy(c);
  return failed ? 1 : 0;
}
