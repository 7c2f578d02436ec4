// lcs_api.hip -- C ABI (include/lcs.h) over the HIP kernels.  Host-side glue only: workspace
// management, H2D/D2H staging for the host-buffer entry points, launch sequencing.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <algorithm>

#include "lcs_internal.h"

namespace {



template <typename T>
int dev_alloc(lcs_ctx *c, T **p, size_t n) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (n == 0) return LCS_OK;
  HIPCHK(c, hipMalloc((void **)p, n * sizeof(T)));
  return LCS_OK;
}

int pinned(lcs_ctx *c, size_t bytes) {
  if (bytes <= c->h_pinned_bytes) return LCS_OK;
  if (c->h_pinned) (void)hipHostFree(c->h_pinned);
  c->h_pinned = nullptr;
  c->h_pinned_bytes = 0;
  HIPCHK(c, hipHostMalloc(&c->h_pinned, bytes, hipHostMallocDefault));
  c->h_pinned_bytes = bytes;
  return LCS_OK;
}

XcGeom make_geo(uint32_t n_cap, int n_f, int ds) {
  XcGeom g;
  g.n_cap = n_cap;
  g.n_f = n_f;
  g.n_tmpl = 3 * n_f;
  g.G = (g.n_tmpl + LCS_TG - 1) / LCS_TG;
  g.n_comb = (int)((n_cap - 136 - 100) / 9600);   // ref src/searcher.cpp:276
  g.ds = ds;
  return g;
}

// Grow the workspace so that n_slots buffers of n_cap samples with n_f hypotheses fit.
int ensure_ws(lcs_ctx *c, int n_slots, uint32_t n_cap, int n_f, bool debug) {
  if (n_slots <= c->cap_slots && n_cap <= c->cap_n_cap && n_f <= c->cap_n_f && (!debug || c->cap_debug)) return LCS_OK;
  n_slots = std::max(n_slots, c->cap_slots);
  n_cap = std::max(n_cap, c->cap_n_cap);
  n_f = std::max(n_f, c->cap_n_f);
  debug = debug || c->cap_debug;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const size_t S = n_slots, NE = 3 * LCS_N_IDX;
  const int G = (3 * n_f + LCS_TG - 1) / LCS_TG;
  int rc;
#define A(p, n) if ((rc = dev_alloc(c, &c->p, (n))) != LCS_OK) return rc
  A(cap32, S * n_cap);
  A(cap64, S * n_cap);
  A(params, S);
  A(fset, (size_t)LCS_NF_MAX);
  A(tmpl, S * LCS_NF_MAX * 3 * 137);
  A(start, S * LCS_NW_MAX * LCS_NF_MAX);
  A(smin, S * LCS_NW_MAX * LCS_G_MAX);
  A(kp2, S * LCS_NW_MAX * LCS_G_MAX);
  A(btab, S * LCS_NW_MAX * G * LCS_KP2_MAX * 64);
  A(single, S * NE * n_f);
  A(pow_, S * NE);
  A(work, S * NE);
  A(frq, S * NE);
  A(spinc, S * LCS_N_IDX);
  A(zth, S * LCS_N_IDX);
  A(peaks, S * LCS_MAXP);
  A(npeaks, S);
  if (debug) {
    A(incoh, S * NE * n_f);
    A(sp, S * LCS_NW_MAX * LCS_N_IDX);
  }
#undef A
  c->cap_slots = n_slots;
  c->cap_n_cap = n_cap;
  c->cap_n_f = n_f;
  c->cap_debug = debug;
  return LCS_OK;
}

// Host-side check that the frequency grid fits the fused combining (see k_prep_tables).
int validate_grid(lcs_ctx *c, const XcGeom &geo, const double *fset, double fc_req, double fc_prog, double fs_prog) {
  for (int w = 0; w < geo.n_comb; ++w) {
    for (int g = 0; g < geo.G; ++g) {
      const int c_hi = std::min(g * LCS_TG + LCS_TG - 1, geo.n_tmpl - 1);
      const int f_lo = (g * LCS_TG) / 3, f_hi = c_hi / 3;
      int mn = 0, mx = 0;
      for (int f = f_lo; f <= f_hi; ++f) {
        const double kf = (fc_req - fset[f]) / fc_prog;
        const int s = (int)std::rint((((double)w * .005) * kf) * fs_prog);
        if (f == f_lo) { mn = mx = s; } else { mn = std::min(mn, s); mx = std::max(mx, s); }
      }
      if ((137 + (mx - mn) + 1) / 2 > LCS_KP2_MAX - LCS_KP2_UNROLL) {
        c->err = "f_search_set too sparse: window-start spread inside one 16-template group exceeds the fused kernel's limit";
        return LCS_ERR_BAD_ARG;
      }
    }
  }
  return LCS_OK;
}

int check_common(lcs_ctx *c, uint32_t n_cap, int n_f) {
  if (!c) return LCS_ERR_BAD_ARG;
  if (n_f < 1 || n_f > LCS_NF_MAX) { c->err = "n_f out of range (1..128)"; return LCS_ERR_BAD_ARG; }
  if (n_cap < 136 + 137 + 9600 + 100) { c->err = "capture buffer shorter than one 5 ms window"; return LCS_ERR_BAD_ARG; }
  if ((n_cap - 136 - 100) / 9600 > LCS_NW_MAX) { c->err = "capture buffer longer than 16 combining windows"; return LCS_ERR_BAD_ARG; }
  return LCS_OK;
}

}  // namespace

extern "C" {

const char *lcs_version(void) { return "lcs_amd 0.1 (gfx950)"; }

void lcs_cell_init(lcs_cell *c) {
  c->fc_requested = NAN; c->fc_programmed = NAN; c->pss_pow = NAN; c->freq = NAN; c->frame_start = NAN;
  c->freq_fine = NAN; c->freq_superfine = NAN; c->ind = -1; c->n_id_2 = -1; c->n_id_1 = -1;
  c->cp_type = LCS_CP_UNKNOWN; c->n_ports = -1; c->n_rb_dl = -1; c->phich_duration = 0; c->phich_resource = 0;
  c->sfn = -1; c->reserved = 0;
}

int lcs_create(int device, lcs_ctx **out) {
  if (!out) return LCS_ERR_BAD_ARG;
  *out = nullptr;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return LCS_ERR_NO_DEVICE;
  if (device < 0) { if (hipGetDevice(&device) != hipSuccess) return LCS_ERR_NO_DEVICE; }
  if (device >= n_dev) return LCS_ERR_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return LCS_ERR_NO_DEVICE;
  lcs_ctx *c = new lcs_ctx();
  c->device = device;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return LCS_ERR_HIP; }
  (void)hipEventCreate(&c->ev_xc0);
  (void)hipEventCreate(&c->ev_xc1);
  // constant tables
  std::vector<double> td(3 * 137 * 2), fd(3 * 62 * 2);
  for (int t = 0; t < 3; ++t) { lcs_tables::pss_td(t, &td[t * 137 * 2]); lcs_tables::pss_fd(t, &fd[t * 62 * 2]); }
  std::vector<int8_t> sss(168 * 3 * 2 * 62);
  for (int n1 = 0; n1 < 168; ++n1)
    for (int n2 = 0; n2 < 3; ++n2)
      for (int s = 0; s < 2; ++s) {
        int32_t tmp[62];
        lcs_tables::sss_fd(n1, n2, s * 10, tmp);
        for (int i = 0; i < 62; ++i) sss[((n1 * 3 + n2) * 2 + s) * 62 + i] = (int8_t)tmp[i];
      }
  std::vector<uint8_t> scr(504 * 1920);
  for (int id = 0; id < 504; ++id) lcs_tables::lte_pn((uint32_t)id, 1920, &scr[(size_t)id * 1920]);
  bool ok = hipMalloc((void **)&c->d_pss_td, td.size() * sizeof(double)) == hipSuccess &&
            hipMalloc((void **)&c->d_pss_fd, fd.size() * sizeof(double)) == hipSuccess &&
            hipMalloc((void **)&c->d_sss_fd, sss.size()) == hipSuccess &&
            hipMalloc((void **)&c->d_pbch_scr, scr.size()) == hipSuccess &&
            hipMemcpy(c->d_pss_td, td.data(), td.size() * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(c->d_pss_fd, fd.data(), fd.size() * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(c->d_sss_fd, sss.data(), sss.size(), hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(c->d_pbch_scr, scr.data(), scr.size(), hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) { lcs_destroy(c); return LCS_ERR_HIP; }
  *out = c;
  return LCS_OK;
}

void lcs_destroy(lcs_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  void *ptrs[] = {c->cap32, c->cap64, c->params, c->fset, c->tmpl, c->start, c->smin, c->kp2, c->btab, c->single,
                  c->incoh, c->pow_, c->work, c->spinc, c->zth, c->sp, c->frq, c->peaks, c->npeaks, c->xc,
                  c->work_items, c->n_work, c->tfg, c->tfg_comp, c->ce, c->tfg_ts, c->tfg_ts_comp, c->cell_scratch,
                  c->cells_out, c->d_pss_td, c->d_pss_fd, c->d_sss_fd, c->d_pbch_scr};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  if (c->h_pinned) (void)hipHostFree(c->h_pinned);
  if (c->ev_xc0) (void)hipEventDestroy(c->ev_xc0);
  if (c->ev_xc1) (void)hipEventDestroy(c->ev_xc1);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char *lcs_last_error(const lcs_ctx *c) { return c ? c->err.c_str() : "null context"; }

int lcs_set_xcorr_variant(lcs_ctx *c, int variant) {
  if (!c || variant < 0 || variant > 1) return LCS_ERR_BAD_ARG;
  c->xcorr_variant = variant;
  return LCS_OK;
}

void *lcs_stream(lcs_ctx *c) { return c ? (void *)c->stream : nullptr; }

int lcs_sync(lcs_ctx *c) {
  if (!c) return LCS_ERR_BAD_ARG;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LCS_OK;
}

// ------------------------------------------------------------------------- xcorr_pss
int lcs_xcorr_pss(lcs_ctx *c, const double *capbuf, uint32_t n_cap, const double *f_search_set, uint16_t n_f,
                  uint8_t ds_comb_arm, double fc_req, double fc_prog, double fs_prog, double *pow_, int32_t *frq,
                  float *single, float *incoh, double *sp_incoherent, float *xc, double *sp, uint16_t *n_comb_xc,
                  uint16_t *n_comb_sp) {
  int rc = check_common(c, n_cap, n_f);
  if (rc) return rc;
  if (!capbuf || !f_search_set || !pow_ || !frq || !single || !sp_incoherent) { c->err = "null argument"; return LCS_ERR_BAD_ARG; }
  HIPCHK(c, hipSetDevice(c->device));
  const bool debug = incoh || sp;
  if ((rc = ensure_ws(c, 1, n_cap, n_f, debug))) return rc;
  const XcGeom geo = make_geo(n_cap, n_f, ds_comb_arm);
  if ((rc = validate_grid(c, geo, f_search_set, fc_req, fc_prog, fs_prog))) return rc;
  SlotParams p{fc_req, fc_prog, fs_prog};
  HIPCHK(c, hipMemcpyAsync(c->cap64, capbuf, sizeof(double2) * n_cap, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->fset, f_search_set, sizeof(double) * n_f, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->params, &p, sizeof(p), hipMemcpyHostToDevice, c->stream));
  if ((rc = lcs_launch_ingest(c, nullptr, 2, 1, n_cap))) return rc;
  double *sp_save = c->sp;
  if (!sp) c->sp = nullptr;   // only materialise sp when asked for
  rc = lcs_launch_xcorr(c, 1, geo, incoh != nullptr, false);
  c->sp = sp_save;
  if (rc) return rc;
  const size_t NE = 3 * LCS_N_IDX;
  HIPCHK(c, hipMemcpyAsync(pow_, c->pow_, sizeof(double) * NE, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(frq, c->frq, sizeof(int) * NE, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(single, c->single, sizeof(float) * NE * n_f, hipMemcpyDeviceToHost, c->stream));
  if (incoh) HIPCHK(c, hipMemcpyAsync(incoh, c->incoh, sizeof(float) * NE * n_f, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(sp_incoherent, c->spinc, sizeof(double) * LCS_N_IDX, hipMemcpyDeviceToHost, c->stream));
  const int ncsp = (int)((n_cap - 136 - 137) / 9600);
  if (sp) HIPCHK(c, hipMemcpyAsync(sp, c->sp, sizeof(double) * ncsp * LCS_N_IDX, hipMemcpyDeviceToHost, c->stream));
  if (xc) {
    const size_t n = 3 * (size_t)(n_cap - 136) * n_f;
    if (n > c->xc_elems) { if ((rc = dev_alloc(c, &c->xc, n))) return rc; c->xc_elems = n; }
    if ((rc = lcs_launch_xc_debug(c, geo))) return rc;
    HIPCHK(c, hipMemcpyAsync(xc, c->xc, sizeof(float2) * n, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (n_comb_xc) *n_comb_xc = (uint16_t)geo.n_comb;
  if (n_comb_sp) *n_comb_sp = (uint16_t)ncsp;
  return LCS_OK;
}

// ------------------------------------------------------------------------ peak_search
int lcs_peak_search(lcs_ctx *c, const double *pow_, const int32_t *frq, const double *Z_th1, const double *f_search_set,
                    uint16_t n_f, double fc_req, double fc_prog, const float *single, uint8_t ds_comb_arm,
                    lcs_cell *cells, int max_cells, int *n_cells) {
  if (!c) return LCS_ERR_BAD_ARG;
  if (n_f < 1 || n_f > LCS_NF_MAX) { c->err = "n_f out of range (1..128)"; return LCS_ERR_BAD_ARG; }
  if (!pow_ || !frq || !Z_th1 || !f_search_set || !single || !n_cells || (max_cells > 0 && !cells)) { c->err = "null argument"; return LCS_ERR_BAD_ARG; }
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  if ((rc = ensure_ws(c, 1, std::max<uint32_t>(c->cap_n_cap, 153600), n_f, false))) return rc;
  const size_t NE = 3 * LCS_N_IDX;
  SlotParams p{fc_req, fc_prog, 0.0};
  HIPCHK(c, hipMemcpyAsync(c->pow_, pow_, sizeof(double) * NE, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->frq, frq, sizeof(int) * NE, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->zth, Z_th1, sizeof(double) * LCS_N_IDX, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->fset, f_search_set, sizeof(double) * n_f, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->single, single, sizeof(float) * NE * n_f, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->params, &p, sizeof(p), hipMemcpyHostToDevice, c->stream));
  XcGeom geo = make_geo(153600, n_f, ds_comb_arm);
  if ((rc = lcs_launch_peak_search(c, 1, geo, std::pow(10.0, -12.0 / 10.0)))) return rc;
  std::vector<lcs_cell> tmp(LCS_MAXP);
  int n = 0;
  HIPCHK(c, hipMemcpyAsync(tmp.data(), c->peaks, sizeof(lcs_cell) * LCS_MAXP, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&n, c->npeaks, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *n_cells = n;
  const int lim = std::min(std::min(n, max_cells), (int)LCS_MAXP);
  for (int i = 0; i < lim; ++i) cells[i] = tmp[i];
  if (n > lim) { c->err = "more peaks than the output array (or LCS_MAXP) holds"; return LCS_ERR_OVERFLOW; }
  return LCS_OK;
}

// ------------------------------------------------------------------- batched chain
int lcs_batch_enqueue(lcs_ctx *c, const void *d_capbufs, int fmt, int n_buf, uint32_t n_cap, const double *f_search_set,
                      uint16_t n_f, const double *fc_requested, const double *fc_programmed, double fs_programmed,
                      int stage_mask) {
  int rc = check_common(c, n_cap, n_f);
  if (rc) return rc;
  if (!d_capbufs || !f_search_set || !fc_requested || !fc_programmed || n_buf < 1) { c->err = "bad argument"; return LCS_ERR_BAD_ARG; }
  if (fmt != LCS_FMT_C64 && fmt != LCS_FMT_IQ_U8) { c->err = "unknown capture format"; return LCS_ERR_BAD_ARG; }
  HIPCHK(c, hipSetDevice(c->device));
  if ((rc = ensure_ws(c, n_buf, n_cap, n_f, false))) return rc;
  const XcGeom geo = make_geo(n_cap, n_f, 2 /* DS_COMB_ARM, ref src/CellSearch.cpp:484 */);
  if ((rc = pinned(c, sizeof(SlotParams) * n_buf + sizeof(double) * LCS_NF_MAX))) return rc;
  SlotParams *hp = (SlotParams *)c->h_pinned;
  double *hf = (double *)(hp + n_buf);
  for (int i = 0; i < n_buf; ++i) {
    hp[i] = SlotParams{fc_requested[i], fc_programmed[i], fs_programmed};
    if (i == 0 || fc_requested[i] != fc_requested[i - 1] || fc_programmed[i] != fc_programmed[i - 1])
      if ((rc = validate_grid(c, geo, f_search_set, fc_requested[i], fc_programmed[i], fs_programmed))) return rc;
  }
  std::memcpy(hf, f_search_set, sizeof(double) * n_f);
  HIPCHK(c, hipMemcpyAsync(c->params, hp, sizeof(SlotParams) * n_buf, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->fset, hf, sizeof(double) * n_f, hipMemcpyHostToDevice, c->stream));
  if ((rc = lcs_launch_ingest(c, d_capbufs, fmt, n_buf, n_cap))) return rc;
  if ((rc = lcs_launch_xcorr(c, n_buf, geo, false, true))) return rc;
  if ((rc = lcs_launch_peak_search(c, n_buf, geo, std::pow(10.0, -12.0 / 10.0)))) return rc;
  if (stage_mask & 2) {
    c->err = "full-chain stages are not built into this library yet";
    return LCS_ERR_BAD_ARG;
  }
  c->last_n_buf = n_buf;
  c->last_stage_mask = stage_mask;
  c->last_geo = geo;
  return LCS_OK;
}

int lcs_batch_collect(lcs_ctx *c, lcs_cell *cells, int max_cells_per_buf, int *n_cells) {
  if (!c || !n_cells || c->last_n_buf <= 0) return LCS_ERR_BAD_ARG;
  const int nb = c->last_n_buf;
  std::vector<lcs_cell> tmp((size_t)nb * LCS_MAXP);
  std::vector<int> cnt(nb);
  HIPCHK(c, hipMemcpyAsync(tmp.data(), c->peaks, sizeof(lcs_cell) * tmp.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(cnt.data(), c->npeaks, sizeof(int) * nb, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int rc = LCS_OK;
  for (int b = 0; b < nb; ++b) {
    const int lim = std::min(std::min(cnt[b], max_cells_per_buf), (int)LCS_MAXP);
    n_cells[b] = cnt[b];
    for (int i = 0; i < lim; ++i) cells[(size_t)b * max_cells_per_buf + i] = tmp[(size_t)b * LCS_MAXP + i];
    if (cnt[b] > lim) rc = LCS_ERR_OVERFLOW;
  }
  if (rc) c->err = "more results than the output array holds";
  return rc;
}

int lcs_search_batch_dev(lcs_ctx *c, const void *d_capbufs, int fmt, int n_buf, uint32_t n_cap, const double *f_search_set,
                         uint16_t n_f, const double *fc_requested, const double *fc_programmed, double fs_programmed,
                         int stage_mask, lcs_cell *cells, int max_cells_per_buf, int *n_cells) {
  int rc = lcs_batch_enqueue(c, d_capbufs, fmt, n_buf, n_cap, f_search_set, n_f, fc_requested, fc_programmed,
                             fs_programmed, stage_mask);
  if (rc) return rc;
  return lcs_batch_collect(c, cells, max_cells_per_buf, n_cells);
}

int lcs_last_xcorr_ms(lcs_ctx *c, float *ms, int *n_launches) {
  if (!c || !ms) return LCS_ERR_BAD_ARG;
  HIPCHK(c, hipEventSynchronize(c->ev_xc1));
  HIPCHK(c, hipEventElapsedTime(ms, c->ev_xc0, c->ev_xc1));
  if (n_launches) *n_launches = c->last_xc_launches;
  return LCS_OK;
}

// ---------------------------------------------------------------------------- tables
int lcs_table_pss_td(int n_id_2, double *out) { if (n_id_2 < 0 || n_id_2 > 2 || !out) return LCS_ERR_BAD_ARG; lcs_tables::pss_td(n_id_2, out); return LCS_OK; }
int lcs_table_pss_fd(int n_id_2, double *out) { if (n_id_2 < 0 || n_id_2 > 2 || !out) return LCS_ERR_BAD_ARG; lcs_tables::pss_fd(n_id_2, out); return LCS_OK; }
int lcs_table_sss_fd(int n_id_1, int n_id_2, int slot_num, int32_t *out) {
  if (n_id_1 < 0 || n_id_1 > 167 || n_id_2 < 0 || n_id_2 > 2 || !out) return LCS_ERR_BAD_ARG;
  lcs_tables::sss_fd(n_id_1, n_id_2, slot_num, out);
  return LCS_OK;
}
int lcs_table_lte_pn(uint32_t c_init, uint32_t len, uint8_t *out) { if (!out) return LCS_ERR_BAD_ARG; lcs_tables::lte_pn(c_init, len, out); return LCS_OK; }
double lcs_chi2cdf_inv(double p, double k) { return lcs_tables::chi2cdf_inv(p, k); }

}  // extern "C"
