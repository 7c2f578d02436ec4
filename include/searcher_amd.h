// searcher_amd.h -- the reference's searcher call surface (include/searcher.h:22-119) on top of
// the C ABI of liblcs_amd.so.  Same function names, argument order and meaning; the only extra
// is the explicit context created once per process/GPU.
//
// Container types come from LCS_CONTAINER_NS (default: lcsc, include/lcs_containers.h).  With
// IT++ installed, compile with -DLCS_CONTAINER_NS=itpp after including <itpp/itbase.h>:
// itpp::Vec/Mat expose the same length()/rows()/cols()/set_size()/_data() members.
//
// Differences to the reference, all forced by the C boundary:
//  * `Cell` is the POD lcs_cell plus the two member functions callers use (n_id_cell, n_symb_dl);
//  * RS_DL is not an argument of tfoec/decode_mib: it is a pure function of the cell identity
//    and is rebuilt on the device (the reference constructs it right before the calls,
//    src/CellSearch.cpp:545);
//  * errors: the reference either succeeds or throws a const char*; these wrappers throw
//    lcs::error carrying the library's message.
#ifndef SEARCHER_AMD_H
#define SEARCHER_AMD_H

#include <cmath>
#include <complex>
#include <list>
#include <stdexcept>
#include <string>
#include <vector>

#include "lcs.h"
#ifndef LCS_CONTAINER_NS
#include "lcs_containers.h"
#define LCS_CONTAINER_NS lcsc
#endif

// include/common.h.in:41-44
typedef std::vector<std::vector<std::vector<std::complex<float> > > > vcf3d;
typedef std::vector<std::vector<std::vector<float> > > vf3d;

namespace lcs {
namespace cn = LCS_CONTAINER_NS;

struct error : std::runtime_error {
  explicit error(const std::string &m) : std::runtime_error(m) {}
};

// class Cell (include/common.h.in:101-129): the POD record with the reference's helpers.
struct Cell : lcs_cell {
  Cell() { lcs_cell_init(this); }
  Cell(const lcs_cell &c) : lcs_cell(c) {}
  int n_id_cell() const { return (n_id_1 >= 0 && n_id_2 >= 0) ? (n_id_2 + 3 * n_id_1) : -1; }   // src/common.cpp:29-31
  int n_symb_dl() const { return cp_type == LCS_CP_NORMAL ? 7 : (cp_type == LCS_CP_EXTENDED ? 6 : -1); }
};

class Searcher {
 public:
  explicit Searcher(int device = -1) : h_(0) {
    const int rc = lcs_create(device, &h_);
    if (rc != LCS_OK) throw error("lcs_create failed (an MI355X is required; there is no CPU fallback)");
  }
  ~Searcher() { lcs_destroy(h_); }
  Searcher(const Searcher &) = delete;               // a context has one owner
  Searcher &operator=(const Searcher &) = delete;
  lcs_ctx *handle() { return h_; }
  // context settings (include/lcs.h): the per-cell stages' footprint limit; device-resident complex<float> batches checked for dongle data
  void set_max_cells_in_flight(int n) { check(lcs_set_max_cells_in_flight(h_, n)); }
  void set_float_batch_probe(bool on) { check(lcs_set_float_batch_probe(h_, on ? 1 : 0)); }

  // include/searcher.h:22-41
  void xcorr_pss(const cn::cvec &capbuf, const cn::vec &f_search_set, unsigned char ds_comb_arm, double fc_requested,
                 double fc_programmed, double fs_programmed, cn::mat &xc_incoherent_collapsed_pow,
                 cn::imat &xc_incoherent_collapsed_frq, vf3d &xc_incoherent_single, vf3d &xc_incoherent,
                 cn::vec &sp_incoherent, vcf3d &xc, cn::vec &sp, unsigned short &n_comb_xc, unsigned short &n_comb_sp,
                 bool want_xc = false) {
    const int n_cap = capbuf.length(), n_f = f_search_set.length();
    std::vector<double> pow_(3 * 9600);
    std::vector<int> frq(3 * 9600);
    std::vector<float> single((size_t)3 * 9600 * n_f), inc((size_t)3 * 9600 * n_f);
    sp_incoherent.set_size(9600);
    const int ncsp = (n_cap - 136 - 137) / 9600;
    sp.set_size(ncsp * 9600);
    std::vector<float> xcbuf;
    if (want_xc) xcbuf.resize((size_t)3 * (n_cap - 136) * n_f * 2);
    check(lcs_xcorr_pss(h_, reinterpret_cast<const double *>(capbuf._data()), (uint32_t)n_cap, f_search_set._data(),
                        (uint16_t)n_f, ds_comb_arm, fc_requested, fc_programmed, fs_programmed, pow_.data(), frq.data(),
                        single.data(), inc.data(), sp_incoherent._data(), want_xc ? xcbuf.data() : 0, sp._data(),
                        &n_comb_xc, &n_comb_sp));
    xc_incoherent_collapsed_pow.set_size(3, 9600);
    xc_incoherent_collapsed_frq.set_size(3, 9600);
    for (int t = 0; t < 3; ++t)
      for (int k = 0; k < 9600; ++k) {
        xc_incoherent_collapsed_pow(t, k) = pow_[t * 9600 + k];
        xc_incoherent_collapsed_frq(t, k) = frq[t * 9600 + k];
      }
    to3d(single, n_f, xc_incoherent_single);
    to3d(inc, n_f, xc_incoherent);
    xc.clear();
    if (want_xc) {
      xc.assign(3, std::vector<std::vector<std::complex<float> > >(n_cap - 136, std::vector<std::complex<float> >(n_f)));
      for (int t = 0; t < 3; ++t)
        for (int k = 0; k < n_cap - 136; ++k)
          for (int f = 0; f < n_f; ++f) {
            const size_t o = (((size_t)t * (n_cap - 136) + k) * n_f + f) * 2;
            xc[t][k][f] = std::complex<float>(xcbuf[o], xcbuf[o + 1]);
          }
    }
  }

  // include/searcher.h:44-56 (appends to `cells`)
  void peak_search(const cn::mat &pow_, const cn::imat &frq, const cn::vec &Z_th1, const cn::vec &f_search_set,
                   double fc_requested, double fc_programmed, const vf3d &xc_incoherent_single,
                   unsigned char ds_comb_arm, std::list<Cell> &cells) {
    const int n_f = f_search_set.length();
    std::vector<double> p(3 * 9600);
    std::vector<int> q(3 * 9600);
    for (int t = 0; t < 3; ++t)
      for (int k = 0; k < 9600; ++k) { p[t * 9600 + k] = pow_(t, k); q[t * 9600 + k] = frq(t, k); }
    std::vector<float> single((size_t)3 * 9600 * n_f);
    for (int t = 0; t < 3; ++t)
      for (int k = 0; k < 9600; ++k)
        for (int f = 0; f < n_f; ++f) single[((size_t)t * 9600 + k) * n_f + f] = xc_incoherent_single[t][k][f];
    std::vector<lcs_cell> out(LCS_MAX_PEAKS);
    int n = 0;
    check(lcs_peak_search(h_, p.data(), q.data(), Z_th1._data(), f_search_set._data(), (uint16_t)n_f, fc_requested,
                          fc_programmed, single.data(), ds_comb_arm, out.data(), (int)out.size(), &n));
    for (int i = 0; i < n; ++i) cells.push_back(Cell(out[i]));
  }

  // include/searcher.h:59-76
  Cell sss_detect(const Cell &cell, const cn::cvec &capbuf, double thresh2_n_sigma, double fc_requested,
                  double fc_programmed, double fs_programmed, cn::vec &sss_h1_np_est, cn::vec &sss_h2_np_est,
                  cn::cvec &sss_h1_nrm_est, cn::cvec &sss_h2_nrm_est, cn::cvec &sss_h1_ext_est,
                  cn::cvec &sss_h2_ext_est, cn::mat &log_lik_nrm, cn::mat &log_lik_ext) {
    sss_h1_np_est.set_size(62); sss_h2_np_est.set_size(62);
    sss_h1_nrm_est.set_size(62); sss_h2_nrm_est.set_size(62); sss_h1_ext_est.set_size(62); sss_h2_ext_est.set_size(62);
    std::vector<double> ln(336), le(336);
    Cell out;
    check(lcs_sss_detect(h_, &cell, reinterpret_cast<const double *>(capbuf._data()), (uint32_t)capbuf.length(),
                         thresh2_n_sigma, fc_requested, fc_programmed, fs_programmed, &out, sss_h1_np_est._data(),
                         sss_h2_np_est._data(), reinterpret_cast<double *>(sss_h1_nrm_est._data()),
                         reinterpret_cast<double *>(sss_h2_nrm_est._data()), reinterpret_cast<double *>(sss_h1_ext_est._data()),
                         reinterpret_cast<double *>(sss_h2_ext_est._data()), ln.data(), le.data()));
    log_lik_nrm.set_size(168, 2); log_lik_ext.set_size(168, 2);
    for (int t = 0; t < 168; ++t)
      for (int c = 0; c < 2; ++c) { log_lik_nrm(t, c) = ln[t * 2 + c]; log_lik_ext(t, c) = le[t * 2 + c]; }
    return out;
  }

  // include/searcher.h:79-85
  Cell pss_sss_foe(const Cell &cell_in, const cn::cvec &capbuf, double fc_requested, double fc_programmed,
                   double fs_programmed) {
    Cell out;
    check(lcs_pss_sss_foe(h_, &cell_in, reinterpret_cast<const double *>(capbuf._data()), (uint32_t)capbuf.length(),
                          fc_requested, fc_programmed, fs_programmed, &out));
    return out;
  }

  // include/searcher.h:88-98
  void extract_tfg(const Cell &cell, const cn::cvec &capbuf_raw, double fc_requested, double fc_programmed,
                   double fs_programmed, cn::cmat &tfg, cn::vec &tfg_timestamp) {
    std::vector<double> g((size_t)LCS_TFG_MAX_OFDM * LCS_TFG_NSC * 2), ts(LCS_TFG_MAX_OFDM);
    int n = 0;
    check(lcs_extract_tfg(h_, &cell, reinterpret_cast<const double *>(capbuf_raw._data()), (uint32_t)capbuf_raw.length(),
                          fc_requested, fc_programmed, fs_programmed, g.data(), ts.data(), &n));
    from_rows(g, n, tfg);
    tfg_timestamp.set_size(n);
    for (int t = 0; t < n; ++t) tfg_timestamp(t) = ts[t];
  }

  // include/searcher.h:101-112 (RS_DL is rebuilt on the device)
  Cell tfoec(const Cell &cell, const cn::cmat &tfg, const cn::vec &tfg_timestamp, double fc_requested,
             double fc_programmed, cn::cmat &tfg_comp, cn::vec &tfg_comp_timestamp) {
    const int n = tfg.rows();
    std::vector<double> g, gc((size_t)n * LCS_TFG_NSC * 2);
    to_rows(tfg, g);
    tfg_comp_timestamp.set_size(n);
    Cell out;
    check(lcs_tfoec(h_, &cell, g.data(), tfg_timestamp._data(), n, fc_requested, fc_programmed, gc.data(),
                    tfg_comp_timestamp._data(), &out));
    from_rows(gc, n, tfg_comp);
    return out;
  }

  // include/searcher.h:115-119
  Cell decode_mib(const Cell &cell, const cn::cmat &tfg) {
    std::vector<double> g;
    to_rows(tfg, g);
    Cell out;
    check(lcs_decode_mib(h_, &cell, g.data(), tfg.rows(), &out));
    return out;
  }

  // whole per-buffer chain of the CLI main loop, device-resident (src/CellSearch.cpp:484-558)
  void search_capbuf(const cn::cvec &capbuf, const cn::vec &f_search_set, double fc_requested, double fc_programmed,
                     double fs_programmed, std::list<Cell> &cells) {
    std::vector<lcs_cell> out(LCS_MAX_PEAKS);
    int n = 0;
    check(lcs_search_capbuf(h_, reinterpret_cast<const double *>(capbuf._data()), (uint32_t)capbuf.length(),
                            f_search_set._data(), (uint16_t)f_search_set.length(), fc_requested, fc_programmed,
                            fs_programmed, out.data(), (int)out.size(), &n, 0, 0, 0));
    for (int i = 0; i < n && i < (int)out.size(); ++i) cells.push_back(Cell(out[i]));
  }

  // A sweep's worth of recorded captures at once (the carrier loop of src/CellSearch.cpp:471-569): n_buf host buffers
  // of n_cap samples each as raw RTL-SDR bytes (LCS_FMT_IQ_U8; captures are exactly (u8-127)/128, src/capbuf.cpp:
  // 172-181) or complex<float> (LCS_FMT_C64), one carrier frequency per buffer.  cells[b] receives buffer b's cells.
  void search_batch_host(const void *h_capbufs, int fmt, int n_buf, uint32_t n_cap, const cn::vec &f_search_set,
                         const std::vector<double> &fc_requested, const std::vector<double> &fc_programmed,
                         double fs_programmed, std::vector<std::list<Cell> > &cells, int max_cells_per_buf = LCS_MAX_PEAKS) {
    std::vector<lcs_cell> out((size_t)n_buf * max_cells_per_buf);
    std::vector<int> cnt(n_buf);
    const int rc = lcs_search_batch_host(h_, h_capbufs, fmt, n_buf, n_cap, f_search_set._data(), (uint16_t)f_search_set.length(),
                                         &fc_requested[0], &fc_programmed[0], fs_programmed, LCS_STAGE_FULL, out.data(),
                                         max_cells_per_buf, cnt.data());
    if (rc != LCS_OK && rc != LCS_ERR_OVERFLOW) check(rc);
    overflowed_ = rc == LCS_ERR_OVERFLOW;
    cells.assign(n_buf, std::list<Cell>());
    for (int b = 0; b < n_buf; ++b)
      for (int i = 0; i < cnt[b] && i < max_cells_per_buf; ++i) cells[b].push_back(Cell(out[(size_t)b * max_cells_per_buf + i]));
  }
  bool last_batch_overflowed() const { return overflowed_; }     // more cells than the arrays hold: results truncated

  // The same in two halves, for drivers that keep several batches in flight (one Searcher per batch in flight, used
  // round-robin): enqueue_batch_host returns once the copy and the kernels are queued, collect_batch waits for them.
  // Buffers from host_alloc are page-locked and DMA'd in place (lcs_host_alloc); any other memory is staged.
  void enqueue_batch_host(const void *h_capbufs, int fmt, int n_buf, uint32_t n_cap, const cn::vec &f_search_set,
                          const std::vector<double> &fc_requested, const std::vector<double> &fc_programmed, double fs_programmed) {
    check(lcs_batch_enqueue_host(h_, h_capbufs, fmt, n_buf, n_cap, f_search_set._data(), (uint16_t)f_search_set.length(),
                                 &fc_requested[0], &fc_programmed[0], fs_programmed, LCS_STAGE_FULL));
    pending_ = n_buf;
  }
  void collect_batch(std::vector<std::list<Cell> > &cells, int max_cells_per_buf = LCS_MAX_PEAKS) {
    const int n_buf = pending_;
    std::vector<lcs_cell> out((size_t)n_buf * max_cells_per_buf);
    std::vector<int> cnt(n_buf);
    const int rc = lcs_batch_collect(h_, out.data(), max_cells_per_buf, cnt.data());
    if (rc != LCS_OK && rc != LCS_ERR_OVERFLOW) check(rc);
    overflowed_ = rc == LCS_ERR_OVERFLOW;
    cells.assign(n_buf, std::list<Cell>());
    for (int b = 0; b < n_buf; ++b)
      for (int i = 0; i < cnt[b] && i < max_cells_per_buf; ++i) cells[b].push_back(Cell(out[(size_t)b * max_cells_per_buf + i]));
    pending_ = 0;
  }
  void *host_alloc(size_t bytes) { void *p = 0; check(lcs_host_alloc(h_, bytes, &p)); return p; }
  void host_free(void *p) { check(lcs_host_free(h_, p)); }
  static int device_count() { return lcs_device_count(); }

  // LO calibration step of LTE-Tracker (src/LTE-Tracker.cpp:565-741) on one recorded buffer: search the
  // +-ppm grid shifted by the current correction (:586), keep the strongest decoded cell (:712-722) and
  // return its residual frequency offset; *correction_residual gets the factor of :724-731.  Returns
  // false when no cell could be decoded (the reference loops on a fresh capture in that case).
  bool kalibrate(const cn::cvec &capbuf, double fc_requested, double fc_programmed, double fs_programmed, double ppm,
                 double correction, Cell &best, double *correction_residual = 0) {
    const int n_extra = (int)std::floor((fc_requested * ppm / 1e6 + 2.5e3) / 5e3);
    cn::vec f_search_set;
    f_search_set.set_size(2 * n_extra + 1);
    for (int i = -n_extra; i <= n_extra; ++i) f_search_set._data()[i + n_extra] = (fc_requested * correction - fc_requested) + 5000.0 * i;
    std::list<Cell> cells;
    search_capbuf(capbuf, f_search_set, fc_requested, fc_programmed, fs_programmed, cells);
    if (cells.empty()) return false;
    bool have = false;
    for (std::list<Cell>::const_iterator it = cells.begin(); it != cells.end(); ++it)
      if (!have || it->pss_pow > best.pss_pow) { best = *it; have = true; }
    const double crystal_freq_actual = fc_programmed - best.freq_superfine;
    if (correction_residual) *correction_residual = (fc_requested / fc_requested * fc_programmed) / crystal_freq_actual;
    return true;
  }

  // Streaming searcher (src/searcher_thread.cpp:83-246): see lcs_stream_* in lcs.h
  void stream_open(int fmt, uint32_t n_cap, double fc_requested, double fc_programmed, double fs_programmed) {
    check(lcs_stream_open(h_, fmt, n_cap, fc_requested, fc_programmed, fs_programmed));
  }
  void stream_push(const void *samples, double f_off, const std::vector<int16_t> &tracked_n_id_cell) {
    check(lcs_stream_push(h_, samples, f_off, tracked_n_id_cell.empty() ? 0 : &tracked_n_id_cell[0], (int)tracked_n_id_cell.size()));
  }
  int stream_collect(std::list<Cell> &new_cells, float *gpu_ms = 0) {   // returns the number of tracked cells seen again
    std::vector<lcs_cell> out(LCS_MAX_CELLS_STREAM);
    int n = 0, dup = 0;
    check(lcs_stream_collect(h_, out.data(), (int)out.size(), &n, &dup, gpu_ms));
    for (int i = 0; i < n && i < (int)out.size(); ++i) new_cells.push_back(Cell(out[i]));
    return dup;
  }
  void stream_close() { check(lcs_stream_close(h_)); }

  // Continuous tracking (lcs_track_stream_block): the next n_sym OFDM symbols of every tracked cell, as the producer thread
  // queues them (td [cell][sym][128], freq_off / frame_timing / late [cell][sym]).  The rows that became computable with
  // this block come back in `out`, under their index in the whole stream (include/lcs.h).
  struct TrackRows {
    int n_cells, max_rs, max_off;
    std::vector<double> meas;            // [cell][4][max_rs][LCS_TRK_MEAS]
    std::vector<int32_t> n_meas;         // [cell][4]
    std::vector<int32_t> mib_ok;         // [cell][max_off]
    std::vector<uint64_t> mib_bits;      // [cell][max_off]
    std::vector<int64_t> mib_from;       // [cell]
    std::vector<int32_t> n_mib;          // [cell]
    const double *meas_rows(int cell, int port) const { return &meas[(((size_t)cell * 4 + port) * max_rs) * LCS_TRK_MEAS]; }
  };
  void track_stream_block(std::vector<lcs_track_cell> &cells, int n_sym, const std::complex<double> *td, const double *freq_off,
                          const double *frame_timing, const double *late, double fc_requested, double fc_programmed,
                          double fs_programmed, TrackRows &out) {
    const int n = (int)cells.size();
    out.n_cells = n;
    out.max_rs = n_sym / 3 + 8;
    out.max_off = n_sym / 120 + 4;
    out.meas.assign((size_t)n * 4 * out.max_rs * LCS_TRK_MEAS, 0.0);
    out.n_meas.assign((size_t)n * 4, 0);
    out.mib_ok.assign((size_t)n * out.max_off, -1);
    out.mib_bits.assign((size_t)n * out.max_off, 0);
    out.mib_from.assign(n, 0);
    out.n_mib.assign(n, 0);
    check(lcs_track_stream_block(h_, cells.data(), n, n_sym, td, freq_off, frame_timing, late, fc_requested, fc_programmed,
                                 fs_programmed, 0, 0, 0, 0, 0, 0, out.meas.data(), 0, 0, out.max_rs, out.n_meas.data(),
                                 out.mib_ok.data(), out.mib_bits.data(), out.max_off, out.mib_from.data(), out.n_mib.data()));
  }
  void track_stream_reset() { check(lcs_track_stream_reset(h_)); }

  // The producer thread's symbol extraction on the device (lcs_track_cut; src/producer_thread.cpp:96-131, 196-246): the symbols of
  // cp_type.size() tracked cells cut out of ONE capture buffer in HBM (d_capbuf: n_cap samples of format fmt, the first with timestamp
  // ts_first) into the device array d_td [cell][n_sym][128] complex<double> -- what lcs_track_block (td_on_device) /
  // lcs_track_stream_block take.  late [cell][n_sym], n_cut [cell] = symbols found before the buffer ends.  state: NULL for a buffer
  // cut from its start; otherwise in = (sym_first, pos_first) per cell, out = where the next buffer of the stream continues.
  struct CutState { std::vector<int64_t> sym, pos; };
  void track_cut(const void *d_capbuf, int fmt, uint32_t n_cap, const std::vector<int32_t> &cp_type, const std::vector<double> &frame_timing,
                 const std::vector<double> &freq_off, double fc_requested, double fc_programmed, double fs_programmed, int n_sym, void *d_td,
                 std::vector<double> &late, std::vector<int32_t> &n_cut, double ts_first = 0.0, CutState *state = 0) {
    const int n = (int)cp_type.size();
    late.assign((size_t)n * n_sym, 0.0);
    n_cut.assign(n, 0);
    std::vector<int64_t> pos_next(n, 0);
    const bool cont = state && (int)state->sym.size() == n && (int)state->pos.size() == n;
    check(lcs_track_cut(h_, d_capbuf, fmt, n_cap, ts_first, n, cp_type.data(), frame_timing.data(), freq_off.data(), cont ? state->sym.data() : 0,
                        cont ? state->pos.data() : 0, fc_requested, fc_programmed, fs_programmed, n_sym, d_td, late.data(), n_cut.data(),
                        pos_next.data()));
    if (state) {
      if (!cont) state->sym.assign(n, 0);
      for (int i = 0; i < n; ++i) state->sym[i] += n_cut[i];
      state->pos = pos_next;
    }
  }

 private:
  enum { LCS_MAX_CELLS_STREAM = 64 };
  void check(int rc) {
    if (rc != LCS_OK) throw error(std::string("liblcs_amd: ") + lcs_last_error(h_));
  }
  static void to3d(const std::vector<float> &flat, int n_f, vf3d &out) {
    out.assign(3, std::vector<std::vector<float> >(9600, std::vector<float>(n_f)));
    for (int t = 0; t < 3; ++t)
      for (int k = 0; k < 9600; ++k)
        for (int f = 0; f < n_f; ++f) out[t][k][f] = flat[((size_t)t * 9600 + k) * n_f + f];
  }
  static void to_rows(const cn::cmat &m, std::vector<double> &g) {   // column-major cmat -> [row][72] interleaved
    g.resize((size_t)m.rows() * m.cols() * 2);
    for (int r = 0; r < m.rows(); ++r)
      for (int c = 0; c < m.cols(); ++c) {
        g[((size_t)r * m.cols() + c) * 2] = m(r, c).real();
        g[((size_t)r * m.cols() + c) * 2 + 1] = m(r, c).imag();
      }
  }
  static void from_rows(const std::vector<double> &g, int n, cn::cmat &m) {
    m.set_size(n, LCS_TFG_NSC);
    for (int r = 0; r < n; ++r)
      for (int c = 0; c < LCS_TFG_NSC; ++c)
        m(r, c) = std::complex<double>(g[((size_t)r * LCS_TFG_NSC + c) * 2], g[((size_t)r * LCS_TFG_NSC + c) * 2 + 1]);
  }
  lcs_ctx *h_;
  bool overflowed_ = false;
  int pending_ = 0;
};

// ---- LTE-Tracker's scalar feedback recurrences (src/tracker_thread.cpp) -----------------------------------------------------
// lcs_track_block / lcs_track_stream_block return, per filtered reference symbol, what the reference's tracker thread feeds
// into its three slow loops; the loops themselves are a handful of flops per symbol and stay on the host.  `meas` rows are
// LCS_TRK_MEAS doubles as lcs.h documents them (0 symbol index, 1 np, 2 tp, 3 sp_raw, 4 sp, 5 frequency_offset +
// residual_f, 6 residual_f_np, 7 frame_timing + delay, 8 delay_np), walked in row order = symbol order of one port.
namespace track {
inline double wrap(double x, double lo, double hi) { return (x - lo) - (hi - lo) * std::floor((x - lo) / (hi - lo)) + lo; }   // include/macros.h:45-53

// do_foe's update of the global frequency offset (:235-242)
inline double fold_frequency_offset(double f, const double *meas, int n_rows) {
  for (int r = 0; r < n_rows; ++r) {
    const double *m = meas + (size_t)r * LCS_TRK_MEAS;
    f = (f * (1 / .000001) + m[5] * (1 / m[6])) / (1 / .000001 + 1 / m[6]);
  }
  return f;
}

// do_toe_v2's update of the frame timing (:283-287)
inline double fold_frame_timing(double t, const double *meas, int n_rows) {
  for (int r = 0; r < n_rows; ++r) {
    const double *m = meas + (size_t)r * LCS_TRK_MEAS;
    double diff = wrap(m[7] - t, -19200.0 / 2, 19200.0 / 2);
    diff = (0 * (1 / .0001) + diff * (1 / m[8])) / (1 / .0001 + 1 / m[8]);
    t = (t + diff) - 19200.0 * std::floor((t + diff) / 19200.0);
  }
  return t;
}

// do_mib_decode's walk over the PBCH fifo (:552-745) when every frame offset has been tried in parallel: mib_ok codes as
// lcs_track_block returns them (3 = CRC and fields match = lock, -1 = not attempted: the walk stops there).  A lock or a
// synchronised failure consumes four frames, an unsynchronised failure one.
struct MibLock {
  double failures;
  bool synchronized;
  int attempts;
  bool dropped;
};
inline MibLock mib_lock_walk(const int32_t *mib_ok, int n, double failures = 0.0, bool synchronized = false, double drop_threshold = 400.0) {
  MibLock s = {failures, synchronized, 0, false};
  for (int o = 0; o < n && mib_ok[o] != -1;) {
    ++s.attempts;
    if (mib_ok[o] == 3) { s.synchronized = true; s.failures = 0.0; o += 4; }
    else if (s.synchronized) { s.failures += 1.0; o += 4; }
    else { s.failures += 0.25; o += 1; }
    if (s.failures >= drop_threshold) { s.dropped = true; break; }
  }
  return s;
}
}  // namespace track

}  // namespace lcs
#endif
