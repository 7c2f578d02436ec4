// lcs_api.hip -- C ABI (include/lcs.h) over the HIP kernels.  Host-side glue only: workspace
// management, H2D/D2H staging for the host-buffer entry points, launch sequencing.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <chrono>

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

XcGeom make_geo(uint32_t n_cap, int n_f, int ds, int cpg = LCS_TG) {
  XcGeom g;
  g.n_cap = n_cap;
  g.n_f = n_f;
  g.n_tmpl = 3 * n_f;
  g.cpg = cpg;
  g.G = (g.n_tmpl + cpg - 1) / cpg;
  g.n_comb = (int)((n_cap - 136 - 100) / 9600);   // ref src/searcher.cpp:276
  g.ds = ds;
  g.foi0 = 0;
  g.n_narrow = (n_f == 1 || cpg == 3) ? g.n_comb : 0;     // one hypothesis per group: no spread at all; otherwise pack_grid looks at the grid
  return g;
}

// Grow the workspace so that n_slots buffers of n_cap samples with n_f hypotheses fit.
int ensure_ws(lcs_ctx *c, int n_slots, uint32_t n_cap, int n_f, bool debug, int G_need = 0) {
  if (G_need <= 0) G_need = (3 * n_f + LCS_TG - 1) / LCS_TG;           // dense packing
  if (n_slots <= c->cap_slots && n_cap <= c->cap_n_cap && n_f <= c->cap_n_f && G_need <= c->cap_G && (!debug || c->cap_debug)) return LCS_OK;
  if (c->st_open) {     // the captured graph of the streaming mode holds the current buffers' addresses
    c->err = "this call needs a larger workspace than the open stream was captured with: lcs_stream_close first";
    return LCS_ERR_BAD_ARG;
  }
  n_slots = std::max(n_slots, c->cap_slots);
  n_cap = std::max(n_cap, c->cap_n_cap);
  n_f = std::max(n_f, c->cap_n_f);
  debug = debug || c->cap_debug;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const size_t S = n_slots, NE = 3 * LCS_N_IDX;
  const int G = std::max(std::max(G_need, c->cap_G), (3 * n_f + LCS_TG - 1) / LCS_TG);
  int rc;
#define A(p, n) if ((rc = dev_alloc(c, &c->p, (n))) != LCS_OK) return rc
  A(cap32, S * n_cap);
  A(cap64, (size_t)n_cap);
  A(params, S);
  // per-hypothesis / per-group tables: sized for the largest grid seen; every call lays its own grid out with its own
  // strides n_f and G (k_prep_tables rebuilds them per call)
  A(fset, (size_t)n_f);
  A(tmpl, S * n_f * 3 * 137);
  A(start, S * LCS_NW_MAX * n_f);
  A(smin, S * LCS_NW_MAX * G);
  A(kp2, S * LCS_NW_MAX * G);
  if (c->btab) { (void)hipFree(c->btab); c->btab = nullptr; }      // fp32 kernel's operand tables (0.5 MB per slot and group): allocated by its first launch (lcs_launch_xcorr)
  c->btab_elems = 0;
  A(single, S * G * LCS_N_IDX * LCS_TG);
  A(sref, NE * n_f);
  A(sp, S * LCS_NW_MAX * LCS_N_IDX);
  A(pow_, S * NE);
  A(work, S * NE);
  A(frq, S * NE);
  A(fix_list, S * NE);
  A(n_fix, 4);
  A(second32, S * NE);
  A(spinc, S * LCS_N_IDX);
  A(zth, S * LCS_N_IDX);
  A(peaks, S * LCS_MAXP);
  A(npeaks, S);
  if (debug) {
    A(incoh, S * NE * n_f);
  }
#undef A
  if (c->cap8) { (void)hipFree(c->cap8); c->cap8 = nullptr; }
  if (c->cap8s) { (void)hipFree(c->cap8s); c->cap8s = nullptr; }
  if (c->brow8) { (void)hipFree(c->brow8); c->brow8 = nullptr; }
  if (c->tq) { (void)hipFree(c->tq); c->tq = nullptr; }
  if (c->tsc) { (void)hipFree(c->tsc); c->tsc = nullptr; }
  c->i8_ready = false;
  for (void **q : {(void **)&c->cap16h, (void **)&c->cap16l, (void **)&c->brow16, (void **)&c->texp16, (void **)&c->tsc16, (void **)&c->xmax16, (void **)&c->xpart16})
    if (*q) { (void)hipFree(*q); *q = nullptr; }
  c->f16_ready = false;
  c->foe_ready = false;      // (a pending lcs_foe_partial result lived in the buffers just replaced)
  c->cap_slots = n_slots;
  c->cap_n_cap = n_cap;
  c->cap_n_f = n_f;
  c->cap_G = G;
  c->cap_debug = debug;
  return LCS_OK;
}

// Buffers of the int8 correlation path (u8 sources), sized like the current workspace.
int ensure_i8(lcs_ctx *c) {
  if (c->i8_ready) return LCS_OK;
  if (c->st_open) { c->err = "int8 buffers cannot be (re)allocated while a stream is open: lcs_stream_close first"; return LCS_ERR_BAD_ARG; }
  const size_t S = (size_t)c->cap_slots;
  const int G = c->cap_G;
  int rc;
  const size_t n8 = S * lcs_cap8_stride(c->cap_n_cap);
  if ((rc = dev_alloc(c, &c->cap8, n8)) || (rc = dev_alloc(c, &c->cap8s, n8))) return rc;
  if ((rc = dev_alloc(c, &c->brow8, S * G * (size_t)LCS_I8_IMG))) return rc;
  if ((rc = dev_alloc(c, &c->tq, S * G * LCS_TG))) return rc;
  if ((rc = dev_alloc(c, &c->tsc, S * G * LCS_TG))) return rc;
  c->i8_ready = true;
  return LCS_OK;
}

// Buffers of the fp16 three-product correlation path (complex<float> sources of the batch entry points).
int ensure_f16(lcs_ctx *c) {
  if (c->f16_ready) return LCS_OK;
  if (c->st_open) { c->err = "fp16 buffers cannot be (re)allocated while a stream is open: lcs_stream_close first"; return LCS_ERR_BAD_ARG; }
  const size_t S = (size_t)c->cap_slots;
  int rc;
  const size_t n16 = S * lcs_cap8_stride(c->cap_n_cap);
  if ((rc = dev_alloc(c, &c->cap16h, n16)) || (rc = dev_alloc(c, &c->cap16l, n16))) return rc;
  if ((rc = dev_alloc(c, &c->brow16, S * c->cap_G * (size_t)LCS_F16_IMG))) return rc;
  if ((rc = dev_alloc(c, &c->texp16, S * c->cap_G * LCS_TG)) || (rc = dev_alloc(c, &c->tsc16, S * c->cap_G * LCS_TG))) return rc;
  if ((rc = dev_alloc(c, &c->xmax16, S)) || (rc = dev_alloc(c, &c->xpart16, S * 128))) return rc;
  c->f16_ready = true;
  return LCS_OK;
}

// Buffers of the per-cell stages, allocated on first use for c->max_work cells (~6 MB each: 3 GB at the default 512) and
// again when the limit was raised since (lcs_set_max_cells_in_flight, or by itself after a batch that carried more cells).
int alloc_percell(lcs_ctx *c, size_t W) {
  int rc;
  const size_t GRID = (size_t)LCS_TFG_ROWS * LCS_TFG_NSC;
  c->percell_ready = false;      // until every buffer exists: a failure part-way leaves a context that allocates again, not one
  c->percell_cap = 0;            // that runs kernels on a null pointer
#define A(p, n) if ((rc = dev_alloc(c, &c->p, (n))) != LCS_OK) return rc
  A(work_items, W);
  A(n_work, 4);
  A(tfg, W * GRID);
  A(tfg_comp, W * GRID);
  A(ce, W * 4 * GRID);
  A(tfg_desc, W * (size_t)LCS_TFG_DESC_BYTES);
  A(tfg_ts, W * LCS_TFG_ROWS);
  A(tfg_ts_comp, W * LCS_TFG_ROWS);
  A(cell_scratch, W * LCS_CELL_SCRATCH);
  A(cells_out, W);
  A(d_dbg, 2048);
#undef A
  c->percell_ready = true;
  c->percell_cap = (int)W;
  return LCS_OK;
}
int ensure_percell(lcs_ctx *c) {
  if (c->percell_ready && c->max_work <= c->percell_cap) return LCS_OK;
  if (c->st_open && c->percell_ready) return LCS_OK;      // the open stream's graph holds these addresses: keep what it was captured with
  const int prev = c->percell_ready ? c->percell_cap : 0;
  if (c->percell_ready) HIPCHK(c, hipStreamSynchronize(c->stream));
  const size_t W = (size_t)std::max(c->max_work, prev);
  int rc = alloc_percell(c, W);
  if (rc != LCS_OK && prev > 0 && W > (size_t)prev) {
    // growing failed (6 MB per cell: 512 -> 1024 cells asks for 3 GB more): back to the capacity that worked, and the limit stays
    // there -- more rounds per batch instead of a failed batch
    (void)hipGetLastError();
    c->max_work = prev;
    c->max_work_pinned = true;
    rc = alloc_percell(c, (size_t)prev);
  }
  return rc;
}

// Device block the results of a batch are compacted into (k_pack_results) and its page-locked mirror, sized for the worst case
// of n_buf buffers (every buffer LCS_MAXP records): allocated when a larger batch arrives, never inside lcs_batch_collect.
int ensure_res_pack(lcs_ctx *c, int n_buf) {
  const size_t need = lcs_pack_rec_offset(n_buf) + (size_t)n_buf * LCS_MAXP * sizeof(lcs_cell);
  if (need <= c->res_pack_bytes) return LCS_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream));      // (the streaming mode's graph does not reference this block: it may grow under an open stream)
  if (c->res_pack) (void)hipFree(c->res_pack);
  if (c->h_res) (void)hipHostFree(c->h_res);
  c->res_pack = c->h_res = nullptr;
  c->res_pack_bytes = 0;
  HIPCHK(c, hipMalloc(&c->res_pack, need));
  HIPCHK(c, hipHostMalloc(&c->h_res, need, hipHostMallocDefault));
  c->res_pack_bytes = need;
  return LCS_OK;
}

// Largest tap count (137 + the spread of the window starts inside one template group, see k_prep_tables) of a frequency
// grid packed `cpg` template columns per group.
int grid_taps(const XcGeom &geo, const double *fset, double fc_req, double fc_prog, double fs_prog) {
  int worst = 137;
  for (int w = 0; w < geo.n_comb; ++w)
    for (int g = 0; g < geo.G; ++g) {
      const int c_hi = std::min(g * geo.cpg + geo.cpg - 1, geo.n_tmpl - 1);
      const int f_lo = (g * geo.cpg) / 3, f_hi = c_hi / 3;
      int mn = 0, mx = 0;
      for (int f = f_lo; f <= f_hi; ++f) {
        const double kf = (fc_req - fset[f]) / fc_prog;
        const int s = (int)std::rint((((double)w * .005) * kf) * fs_prog);
        if (f == f_lo) { mn = mx = s; } else { mn = std::min(mn, s); mx = std::max(mx, s); }
      }
      worst = std::max(worst, 137 + (mx - mn));
    }
  return worst;
}

// Number of leading combining windows in which the window starts of every group's hypotheses stay within LCS_NARROW_SPREAD
// samples (the spread grows with the window index; the count stops at the first window that exceeds it).
int grid_narrow_windows(const XcGeom &geo, const double *fset, double fc_req, double fc_prog, double fs_prog) {
  for (int w = 0; w < geo.n_comb; ++w)
    for (int g = 0; g < geo.G; ++g) {
      const int c_hi = std::min(g * geo.cpg + geo.cpg - 1, geo.n_tmpl - 1);
      const int f_lo = (g * geo.cpg) / 3, f_hi = c_hi / 3;
      int mn = 0, mx = 0;
      for (int f = f_lo; f <= f_hi; ++f) {
        const double kf = (fc_req - fset[f]) / fc_prog;
        const int s = (int)std::rint((((double)w * .005) * kf) * fs_prog);
        if (f == f_lo) { mn = mx = s; } else { mn = std::min(mn, s); mx = std::max(mx, s); }
      }
      if (mx - mn > LCS_NARROW_SPREAD) return w;
    }
  return geo.n_comb;
}

// Choose how the 3 n_f templates are packed into 16-column groups: densely when the window starts of a group's
// hypotheses stay within `max_taps` - 137 samples of each other over the whole buffer (every grid the CLI builds), else
// with fewer whole hypotheses per group -- one per group always fits (its three templates share a window start).
XcGeom pack_grid(uint32_t n_cap, int n_f, int ds, const double *fset, const double *fc_req, const double *fc_prog, int n_buf,
                 double fs_prog, int max_taps) {
  static const int packings[] = {LCS_TG, 15, 12, 9, 6, 3};
  for (int cpg : packings) {
    XcGeom geo = make_geo(n_cap, n_f, ds, cpg);
    bool fits = true;
    for (int i = 0; i < n_buf && fits; ++i)
      if (i == 0 || fc_req[i] != fc_req[i - 1] || fc_prog[i] != fc_prog[i - 1])
        fits = grid_taps(geo, fset, fc_req[i], fc_prog[i], fs_prog) <= max_taps;
    if (fits || cpg == 3) {
      int nn = geo.n_comb;
      for (int i = 0; i < n_buf && nn > 0; ++i)
        if (i == 0 || fc_req[i] != fc_req[i - 1] || fc_prog[i] != fc_prog[i - 1])
          nn = std::min(nn, grid_narrow_windows(geo, fset, fc_req[i], fc_prog[i], fs_prog));
      geo.n_narrow = nn;
      return geo;
    }
  }
  return make_geo(n_cap, n_f, ds, 3);
}
constexpr int kMaxTapsI8 = LCS_I8_MAX_TAPS;                               // int8 kernel: 137 taps + delays below LCS_I8_OFF (the fp16 kernel holds 160)
constexpr int kMaxTapsF32 = 2 * (LCS_KP2_MAX - LCS_KP2_UNROLL);          // fp32 kernel: 124 tap pairs

// lcs_set_float_batch_probe: is a batch of complex<float> buffers dongle data -- every component exactly (u8 - 127) / 128 (ref
// src/capbuf.cpp:172-181)?  One pass: the byte each component would have come from, and whether it reproduces the component exactly
// (x 128 + 127 is exact in fp32 for such values; NaN, infinities and anything off the 8-bit grid fail).  The bytes are then handed to
// the u8 route unchanged (int8 copies, int8 correlation kernel, the fp64 stages on the int8 pairs): the same numbers, the faster kernel.
__global__ __launch_bounds__(256) void k_c64_probe_u8(const float *__restrict__ src, size_t n_comp, uint8_t *__restrict__ dst, int *__restrict__ flag) {
  bool ok = true;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n_comp; i += (size_t)gridDim.x * blockDim.x * 4) {
    const float4 x = *reinterpret_cast<const float4 *>(src + i);      // (n_comp = 2 n_cap n_buf is a multiple of 4 for the batch shapes the library takes: checked by the caller)
    const float v[4] = {x.x * 128.0f + 127.0f, x.y * 128.0f + 127.0f, x.z * 128.0f + 127.0f, x.w * 128.0f + 127.0f};
    uchar4 b;
    unsigned char *pb = reinterpret_cast<unsigned char *>(&b);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float r = rintf(v[q]);
      ok = ok && v[q] == r && r >= 0.0f && r <= 255.0f;
      pb[q] = (unsigned char)(int)fminf(fmaxf(r, 0.0f), 255.0f);
    }
    *reinterpret_cast<uchar4 *>(dst + i) = b;
  }
  if (__any(!ok) && (threadIdx.x & 63) == 0) atomicAnd(flag, 0);
}

// complex<double> host buffer -> device (cap64, slot 0) and the choice of the correlation kernel: a buffer whose every
// component is exactly (u8 - 127) / 128 -- any dongle capture -- takes the int8 kernel, anything else the fp32 one.
// Returns the geometry to correlate with; c->use_i8 is set accordingly.
int upload_host_capbuf(lcs_ctx *c, const double *capbuf, uint32_t n_cap, const double *f_search_set, int n_f, int ds, double fc_req,
                       double fc_prog, double fs_prog, bool debug, XcGeom *geo_out) {
  const XcGeom geo32 = pack_grid(n_cap, n_f, ds, f_search_set, &fc_req, &fc_prog, 1, fs_prog, kMaxTapsF32);
  const XcGeom geo8 = pack_grid(n_cap, n_f, ds, f_search_set, &fc_req, &fc_prog, 1, fs_prog, kMaxTapsI8);
  int rc;
  c->foe_ready = false;      // slot 0 is overwritten: a pending lcs_foe_partial result is gone (lcs_foe_partial sets it again)
  if ((rc = ensure_ws(c, 1, n_cap, n_f, debug, std::max(geo32.G, geo8.G)))) return rc;
  // the int8 copies cannot be (re)allocated under an open stream's graph: such a context keeps the fp32 kernel
  const bool can_i8 = c->i8_ready || !c->st_open;
  if (can_i8 && (rc = ensure_i8(c))) return rc;
  c->h_params = SlotParams{fc_req, fc_prog, fs_prog};
  HIPCHK(c, hipMemcpyAsync(c->cap64, capbuf, sizeof(double2) * n_cap, hipMemcpyHostToDevice, c->stream));
  c->cap64_valid = true;
  HIPCHK(c, hipMemcpyAsync(c->fset, f_search_set, sizeof(double) * n_f, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->params, &c->h_params, sizeof(SlotParams), hipMemcpyHostToDevice, c->stream));
  bool exact = false;
  if (can_i8) { if ((rc = lcs_launch_ingest_c128(c, n_cap, &exact))) return rc; }
  else if ((rc = lcs_launch_ingest(c, nullptr, 2, 1, n_cap))) return rc;
  c->use_i8 = exact;
  c->use_f16 = false;
  *geo_out = exact ? geo8 : geo32;
  return LCS_OK;
}

int check_common(lcs_ctx *c, uint32_t n_cap, int n_f) {
  if (!c) return LCS_ERR_BAD_ARG;
  if (n_f < 1 || n_f > LCS_NF_LIMIT) { c->err = "n_f out of range (1..1024)"; return LCS_ERR_BAD_ARG; }
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
  // Two streams per context: the correlation kernel goes to a LOW-priority stream, everything
  // else (small, latency-bound kernels) to a HIGH-priority one, so that when two contexts are
  // used round-robin the tail of batch i is not starved by the correlation of batch i+1.
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  bool ok_streams = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_greatest) == hipSuccess;
  ok_streams = ok_streams && hipStreamCreateWithPriority(&c->stream_xc, hipStreamNonBlocking, prio_least) == hipSuccess;
  if (!ok_streams) { lcs_destroy(c); return LCS_ERR_HIP; }
  // events between streams of ONE device (and the two that time the correlation): no system-scope fence -- its cache
  // write-back and invalidation at every record is paid by the kernels that follow (results reach the host through
  // hipMemcpyAsync / stream synchronisation, which fence on their own)
  (void)hipEventCreateWithFlags(&c->ev_xc0, LCS_EVENT_NOFENCE);
  (void)hipEventCreateWithFlags(&c->ev_xc1, LCS_EVENT_NOFENCE);
  (void)hipEventCreateWithFlags(&c->ev_pre, hipEventDisableTiming | LCS_EVENT_NOFENCE);
  (void)hipEventCreateWithFlags(&c->ev_post, hipEventDisableTiming | LCS_EVENT_NOFENCE);
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
  std::vector<int16_t> derm(2 * 120 * 16, (int16_t)-1);
  for (int v = 0; v < 2; ++v) {                            // 0: normal CP (1920 bits), 1: extended CP (1728)
    const int n_e = v ? 1728 : 1920;
    std::vector<uint8_t> map(n_e);
    lcs_tables::pbch_deratematch_map(n_e, map.data());
    int fill[120] = {0};
    for (int t = 0; t < n_e; ++t) derm[(v * 120 + map[t]) * 16 + fill[map[t]]++] = (int16_t)t;
  }
  uint32_t pn_jump[32];
  lcs_tables::pn_jump_table(1600 + 2 * (110 - 6), pn_jump);
  bool ok = hipMalloc((void **)&c->d_pss_td, td.size() * sizeof(double)) == hipSuccess &&
            hipMalloc((void **)&c->d_flag, sizeof(int)) == hipSuccess &&
            hipMalloc((void **)&c->d_pn_jump, sizeof(pn_jump)) == hipSuccess &&
            hipMemcpy(c->d_pn_jump, pn_jump, sizeof(pn_jump), hipMemcpyHostToDevice) == hipSuccess &&
            hipMalloc((void **)&c->d_pss_fd, fd.size() * sizeof(double)) == hipSuccess &&
            hipMalloc((void **)&c->d_sss_fd, sss.size()) == hipSuccess &&
            hipMalloc((void **)&c->d_pbch_scr, scr.size()) == hipSuccess &&
            hipMalloc((void **)&c->d_derm_inv, derm.size() * sizeof(int16_t)) == hipSuccess &&
            hipMemcpy(c->d_derm_inv, derm.data(), derm.size() * sizeof(int16_t), hipMemcpyHostToDevice) == hipSuccess &&
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
  if (c->st_open) (void)lcs_stream_close(c);
  lcs_track_stream_free(c);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->stream_xc) (void)hipStreamSynchronize(c->stream_xc);
  void *ptrs[] = {c->cap32, c->cap64, c->params, c->fset, c->tmpl, c->start, c->smin, c->kp2, c->btab, c->single,
                  c->incoh, c->sref, c->pow_, c->work, c->spinc, c->zth, c->sp, c->frq, c->fix_list, c->n_fix, c->second32, c->fset_g, c->peaks, c->npeaks, c->xc,
                  c->work_items, c->n_work, c->tfg, c->tfg_comp, c->ce, c->tfg_ts, c->tfg_desc, c->tfg_ts_comp, c->cell_scratch,
                  c->cells_out, c->d_pss_td, c->d_pss_fd, c->d_sss_fd, c->d_pbch_scr, c->d_derm_inv, c->d_dbg, c->pk_items, c->n_pk,
                  c->sss_ws, c->d_pn_jump, c->cap8, c->cap8s, c->brow8, c->tq, c->tsc, c->cap16h, c->cap16l, c->brow16, c->texp16, c->tsc16, c->xmax16, c->xpart16, c->h2d, c->trk_td, c->trk_syms, c->trk_raw, c->trk_ce,
                  c->trk_meta, c->trk_rs, c->trk_fmeta, c->trk_pw, c->trk_idx, c->trk_small, c->trk_cells, c->trk_acfd, c->trk_actd,
                  c->trk_syncce, c->trk_sync, c->d_flag, c->trk_cut_hit, c->trk_cut_meta, c->c64_u8};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  if (c->h_pinned) (void)hipHostFree(c->h_pinned);
  if (c->res_pack) (void)hipFree(c->res_pack);
  if (c->h_res) (void)hipHostFree(c->h_res);
  if (c->trk_hpin) free(c->trk_hpin);
  for (int k = 0; k < 2; ++k) {
    if (c->h_stage[k]) (void)hipHostFree(c->h_stage[k]);
    if (c->ev_stage[k]) (void)hipEventDestroy(c->ev_stage[k]);
  }
  if (c->ev_xc0) (void)hipEventDestroy(c->ev_xc0);
  if (c->ev_xc1) (void)hipEventDestroy(c->ev_xc1);
  if (c->ev_pre) (void)hipEventDestroy(c->ev_pre);
  if (c->ev_post) (void)hipEventDestroy(c->ev_post);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->stream_xc) (void)hipStreamDestroy(c->stream_xc);
  delete c;
}

const char *lcs_last_error(const lcs_ctx *c) { return c ? c->err.c_str() : "null context"; }

int lcs_set_float_batch_probe(lcs_ctx *c, int on) {
  if (!c) return LCS_ERR_BAD_ARG;
  c->c64_probe = on != 0;
  c->c64_skip = 0;
  return LCS_OK;
}

int lcs_set_max_cells_in_flight(lcs_ctx *c, int n) {
  if (!c || n < 1) return LCS_ERR_BAD_ARG;
  c->max_work = std::min(n, (int)LCS_MAX_WORK);
  c->max_work_pinned = true;          // the limit no longer grows by itself
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
  if (ds_comb_arm > 8) { c->err = "ds_comb_arm > 8 is not supported (the reference uses 2)"; return LCS_ERR_BAD_ARG; }
  HIPCHK(c, hipSetDevice(c->device));
  const bool debug = incoh != nullptr;
  XcGeom geo;
  if ((rc = upload_host_capbuf(c, capbuf, n_cap, f_search_set, n_f, ds_comb_arm, fc_req, fc_prog, fs_prog, debug, &geo))) return rc;
  if ((rc = lcs_launch_xcorr(c, 1, geo, incoh != nullptr, false))) return rc;
  if ((rc = lcs_launch_single_layout(c, geo, 0, c->sref, 1))) return rc;
  const size_t NE = 3 * LCS_N_IDX;
  HIPCHK(c, hipMemcpyAsync(pow_, c->pow_, sizeof(double) * NE, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(frq, c->frq, sizeof(int) * NE, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(single, c->sref, sizeof(float) * NE * n_f, hipMemcpyDeviceToHost, c->stream));
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
  if (n_f < 1 || n_f > LCS_NF_LIMIT) { c->err = "n_f out of range (1..1024)"; return LCS_ERR_BAD_ARG; }
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
  HIPCHK(c, hipMemcpyAsync(c->sref, single, sizeof(float) * NE * n_f, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->params, &p, sizeof(p), hipMemcpyHostToDevice, c->stream));
  XcGeom geo = make_geo(153600, n_f, ds_comb_arm);
  if ((rc = lcs_launch_single_layout(c, geo, 0, c->sref, 0))) return rc;
  if ((rc = lcs_launch_peak_search(c, 1, geo, std::pow(10.0, -12.0 / 10.0), false))) return rc;
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
static int percell_round(lcs_ctx *c, int n_buf, uint32_t n_cap, int r) {
  int rc;
  if ((rc = lcs_launch_gather_work(c, n_buf, r * c->round_cells, c->round_cells))) return rc;
  if ((rc = lcs_launch_tfg(c, n_cap, true))) return rc;
  if ((rc = lcs_launch_tfoec(c, false, 2))) return rc;
  return lcs_launch_mib(c, true);
}

int lcs_batch_enqueue(lcs_ctx *c, const void *d_capbufs, int fmt, int n_buf, uint32_t n_cap, const double *f_search_set,
                      uint16_t n_f, const double *fc_requested, const double *fc_programmed, double fs_programmed,
                      int stage_mask) {
  int rc = check_common(c, n_cap, n_f);
  if (rc) return rc;
  if (!d_capbufs || !f_search_set || !fc_requested || !fc_programmed || n_buf < 1) { c->err = "bad argument"; return LCS_ERR_BAD_ARG; }
  if (fmt != LCS_FMT_C64 && fmt != LCS_FMT_IQ_U8) { c->err = "unknown capture format"; return LCS_ERR_BAD_ARG; }
  HIPCHK(c, hipSetDevice(c->device));
  c->foe_ready = false;      // the batch overwrites the buffers a pending lcs_foe_partial left for lcs_foe_finish
  // u8 I/Q is exact in int8: the int8 three-digit kernel (pss_xcorr_i8.hip), 137 taps + window-start spread <= 152 inside
  // every template group; every other source takes the fp32 kernel (spread <= 111); pack_grid thins the groups of a grid
  // that is too sparse for that
  const XcGeom geo = pack_grid(n_cap, n_f, 2 /* DS_COMB_ARM, ref src/CellSearch.cpp:484 */, f_search_set, fc_requested, fc_programmed,
                               n_buf, fs_programmed, kMaxTapsI8);      // the fp16 kernel holds 160 taps per group and shares the int8 kernel's packing
  if ((rc = ensure_ws(c, n_buf, n_cap, n_f, false, geo.G))) return rc;
  if ((rc = ensure_res_pack(c, n_buf))) return rc;
  if ((rc = pinned(c, sizeof(SlotParams) * n_buf + sizeof(double) * n_f))) return rc;
  SlotParams *hp = (SlotParams *)c->h_pinned;
  double *hf = (double *)(hp + n_buf);
  for (int i = 0; i < n_buf; ++i) hp[i] = SlotParams{fc_requested[i], fc_programmed[i], fs_programmed};
  std::memcpy(hf, f_search_set, sizeof(double) * n_f);
  HIPCHK(c, hipMemcpyAsync(c->params, hp, sizeof(SlotParams) * n_buf, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->fset, hf, sizeof(double) * n_f, hipMemcpyHostToDevice, c->stream));
  c->cap64_valid = false;
  // lcs_set_float_batch_probe: a complex<float> batch that is dongle data becomes the u8 batch it came from (one pass + one small
  // read-back: the host waits for it -- while the other contexts' kernels keep the GPU busy -- before it knows which kernels to queue)
  c->last_c64_routed = false;
  if (fmt == LCS_FMT_C64 && c->c64_probe && (c->i8_ready || !c->st_open) && ((size_t)n_buf * n_cap) % 2 == 0 && (reinterpret_cast<uintptr_t>(d_capbufs) & 15) == 0) {
    if (c->c64_skip > 0) --c->c64_skip;
    else {
      const size_t n_comp = (size_t)2 * n_cap * n_buf;
      if (n_comp > c->c64_u8_bytes) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->c64_u8) (void)hipFree(c->c64_u8);
        c->c64_u8 = nullptr; c->c64_u8_bytes = 0;
        HIPCHK(c, hipMalloc((void **)&c->c64_u8, n_comp));
        c->c64_u8_bytes = n_comp;
      }
      int flag = 1;
      HIPCHK(c, hipMemcpyAsync(c->d_flag, &flag, sizeof(int), hipMemcpyHostToDevice, c->stream));
      hipLaunchKernelGGL(k_c64_probe_u8, dim3(2048), dim3(256), 0, c->stream, (const float *)d_capbufs, n_comp, c->c64_u8, c->d_flag);
      HIPCHK(c, hipMemcpyAsync(&flag, c->d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      if (flag) { d_capbufs = c->c64_u8; fmt = LCS_FMT_IQ_U8; c->last_c64_routed = true; }
      else c->c64_skip = 15;      // a float front end: the next batches are not probed (one in sixteen is)
    }
  }
  const int fmt_in = c->last_c64_routed ? LCS_FMT_C64 : fmt;      // what the caller handed over: the hint bookkeeping goes by it
  c->use_i8 = fmt == LCS_FMT_IQ_U8;
  // complex<float> sources: fp16 hi / lo operands, three products (pss_xcorr_f16.hip) -- unless its buffers would have to be
  // allocated under an open stream's graph: such a context keeps the fp32 kernel for them (160 taps per group fit it too)
  c->use_f16 = fmt == LCS_FMT_C64 && (c->f16_ready || !c->st_open);
  if (fmt == LCS_FMT_IQ_U8 && (rc = ensure_i8(c))) return rc;      // int8 copies: every u8 source (the fp64 stages read them)
  if (c->use_f16 && (rc = ensure_f16(c))) return rc;
  if (c->use_f16) { if ((rc = lcs_launch_ingest_f16(c, d_capbufs, n_buf, n_cap))) return rc; }
  else if ((rc = lcs_launch_ingest(c, d_capbufs, fmt, n_buf, n_cap))) return rc;
  if ((rc = lcs_launch_xcorr(c, n_buf, geo, false, true))) return rc;
  if ((rc = lcs_launch_peak_search(c, n_buf, geo, std::pow(10.0, -12.0 / 10.0), true))) return rc;
  if (stage_mask & 2) {
    // The per-cell stages hold max_work cells at a time, in rounds.  Round 4: the batch before is the predictor for how many
    // rounds to enqueue and how wide the per-cell grids are.  A busy band carries 4-5 cells per buffer past SSS: with the
    // rounds and grids sized for one cell per buffer, every such batch needed a second round launched from
    // lcs_batch_collect (a host round trip in the middle of the pipeline) and every workgroup walked ~8 cells one after
    // the other.  A wrong guess costs time only: the kernels loop over whatever the list holds, and collect launches the
    // missing rounds, so no batch overflows and sparse batches pay for no empty rounds.  The hint was measured on a batch
    // of hint_n_buf buffers of one format and stage mask: it is scaled to this batch's size and forgotten when the shape changed.
    int hint = 0;
    if (c->hint_n_buf > 0 && c->hint_fmt == fmt_in && c->hint_stage == stage_mask)
      hint = (int)std::min<long long>((long long)c->work_hint * n_buf / c->hint_n_buf, (long long)n_buf * LCS_MAXP);
    // a batch that carried more cells than a round holds: the limit doubles (up to LCS_MAX_WORK) unless the caller pinned it
    if (!c->max_work_pinned && !c->st_open && hint > c->max_work && c->max_work < LCS_MAX_WORK)
      c->max_work = std::min<int>(LCS_MAX_WORK, 2 * c->max_work);
    if ((rc = ensure_percell(c))) return rc;
    if ((rc = lcs_launch_sss_foe(c, n_buf, n_cap, 3.0 /* THRESH2_N_SIGMA, ref src/CellSearch.cpp:528 */, nullptr))) return rc;
    c->needed_rows_only = true;
    c->round_cells = std::min(c->max_work, c->percell_cap);      // fixed for this batch
    const int expect = std::max(n_buf, hint + hint / 4);
    c->grid_items = std::min(c->round_cells, std::max(64, std::max(n_buf / 2, hint + hint / 8)));
    const int rounds = std::min(8, (expect + c->round_cells - 1) / c->round_cells);
    for (int r = 0; r < rounds; ++r)
      if ((rc = percell_round(c, n_buf, n_cap, r))) return rc;
    c->last_cell_rounds = rounds;
  }
  if ((rc = lcs_launch_pack_results(c, n_buf, (stage_mask & 2) != 0))) return rc;
  c->last_n_buf = n_buf;
  c->last_stage_mask = stage_mask;
  c->last_fmt = fmt_in;
  c->last_geo = geo;
  return LCS_OK;
}

// Round 5: the results arrive compacted (k_pack_results at the end of the enqueued chain).  ONE copy of the header, the
// per-buffer counts and as many records as the previous batch returned (+ 25 %) into page-locked memory owned by the
// context, one synchronisation; a second copy only when the batch returned more.  No allocation, no pageable staging
// (rounds 1-4 copied n_buf x 64 records, 786 KB per 128-buffer batch, into a vector built inside the call).
int lcs_batch_collect(lcs_ctx *c, lcs_cell *cells, int max_cells_per_buf, int *n_cells) {
  if (!c || !n_cells || c->last_n_buf <= 0 || max_cells_per_buf < 0 || (!cells && max_cells_per_buf > 0)) return LCS_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  const int nb = c->last_n_buf;
  const bool full = (c->last_stage_mask & 2) != 0;
  const size_t rec_off = lcs_pack_rec_offset(nb);
  const int *hdr = static_cast<const int *>(c->h_res);
  const int *cnt = hdr + 8;
  const lcs_cell *rec = reinterpret_cast<const lcs_cell *>(static_cast<const char *>(c->h_res) + rec_off);
  int rc = LCS_OK;
  double host_us = 0;                                   // host time of this call outside the wait for the GPU (lcs_last_collect_host_us)
  auto t_sync_done = std::chrono::steady_clock::now();
  for (int pass = 0;; ++pass) {
    const size_t first = std::min<size_t>((size_t)nb * LCS_MAXP, (size_t)std::max(nb / 2, c->collect_hint + c->collect_hint / 4 + 8));
    const auto t_a = std::chrono::steady_clock::now();
    HIPCHK(c, hipMemcpyAsync(c->h_res, c->res_pack, rec_off + first * sizeof(lcs_cell), hipMemcpyDeviceToHost, c->stream));
    const auto t_b = std::chrono::steady_clock::now();
    HIPCHK(c, hipStreamSynchronize(c->stream));
    t_sync_done = std::chrono::steady_clock::now();
    host_us += std::chrono::duration<double, std::micro>(t_b - t_a).count();
    const int work_total = hdr[5];
    if (full && pass == 0) { c->work_hint = work_total; c->hint_n_buf = nb; c->hint_fmt = c->last_fmt; c->hint_stage = c->last_stage_mask; }
    if (full && pass == 0 && work_total > c->last_cell_rounds * c->round_cells) {
      // more cells passed SSS than the enqueued rounds decode: run the remaining rounds now (rare: the first dense batch)
      const int rounds = (work_total + c->round_cells - 1) / c->round_cells;
      for (int r = c->last_cell_rounds; r < rounds; ++r)
        if ((rc = percell_round(c, nb, c->last_geo.n_cap, r))) return rc;
      c->last_cell_rounds = rounds;
      if ((rc = lcs_launch_pack_results(c, nb, true))) return rc;
      continue;
    }
    const int total = hdr[0];
    if ((size_t)total > first) {
      HIPCHK(c, hipMemcpyAsync(static_cast<char *>(c->h_res) + rec_off + first * sizeof(lcs_cell),
                               static_cast<const char *>(c->res_pack) + rec_off + first * sizeof(lcs_cell),
                               ((size_t)total - first) * sizeof(lcs_cell), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    c->collect_hint = total;
    break;
  }
  if (hdr[1]) rc = LCS_ERR_OVERFLOW;          // a buffer with more than LCS_MAXP peaks: only with non-positive thresholds (lcs.h)
  size_t at = 0;
  for (int b = 0; b < nb; ++b) {
    const int n = cnt[b];
    const int take = std::min(n, max_cells_per_buf);
    if (take > 0) std::memcpy(cells + (size_t)b * max_cells_per_buf, rec + at, (size_t)take * sizeof(lcs_cell));
    if (n > max_cells_per_buf) rc = LCS_ERR_OVERFLOW;
    at += (size_t)n;
    n_cells[b] = n;
  }
  c->src32 = nullptr;      // complex<float> batches were read in place: the caller's buffers are no longer referenced
  c->last_collect_host_us = host_us + std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_sync_done).count();
  if (rc) c->err = "more results than the output array holds";
  return rc;
}

int lcs_last_batch_stats(lcs_ctx *c, int stats[8]) {
  if (!c || !stats || !c->h_res || c->last_n_buf <= 0) return LCS_ERR_BAD_ARG;
  std::memcpy(stats, c->h_res, 8 * sizeof(int));      // the header lcs_batch_collect brought over (k_pack_results)
  return LCS_OK;
}

int lcs_last_collect_host_us(lcs_ctx *c, double *us) {
  if (!c || !us) return LCS_ERR_BAD_ARG;
  *us = c->last_collect_host_us;
  return LCS_OK;
}

int lcs_search_batch_dev(lcs_ctx *c, const void *d_capbufs, int fmt, int n_buf, uint32_t n_cap, const double *f_search_set,
                         uint16_t n_f, const double *fc_requested, const double *fc_programmed, double fs_programmed,
                         int stage_mask, lcs_cell *cells, int max_cells_per_buf, int *n_cells) {
  int rc = lcs_batch_enqueue(c, d_capbufs, fmt, n_buf, n_cap, f_search_set, n_f, fc_requested, fc_programmed,
                             fs_programmed, stage_mask);
  if (rc) return rc;
  return lcs_batch_collect(c, cells, max_cells_per_buf, n_cells);
}

// ---- host-fed batches ---------------------------------------------------------------------------------------------
// What a caller holding recorded capbuf_NNNN.it files or dongle bytes uses (the carrier loop of src/CellSearch.cpp:471-569
// with the captures in host memory).  The H2D copy is asynchronous on the context's stream: with two or three contexts
// used round-robin (enqueue batch i + 1, then collect batch i) the PCIe transfer of one batch runs under the kernels of
// the previous one.  That needs page-locked source memory: buffers from lcs_host_alloc are DMA'd in place; any other
// pointer is staged through two pinned 4 MB slots owned by the context (CPU memcpy of chunk k + 1 under the DMA of
// chunk k -- correct for every pointer, but then the CPU copy, ~10 GB/s, is what bounds the transfer).
int lcs_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int lcs_host_alloc(lcs_ctx *c, size_t bytes, void **out) {
  if (!c || !out) return LCS_ERR_BAD_ARG;
  *out = nullptr;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
  return LCS_OK;
}

int lcs_host_free(lcs_ctx *c, void *p) {
  if (!c) return LCS_ERR_BAD_ARG;
  if (p) HIPCHK(c, hipHostFree(p));
  return LCS_OK;
}

// Device memory for callers without a HIP toolchain of their own (the host tools are plain g++): buffers they hand to the
// device-resident entry points (lcs_batch_enqueue, lcs_track_block with td_on_device, lcs_track_stream_block).
int lcs_device_alloc(lcs_ctx *c, size_t bytes, void **out) {
  if (!c || !out) return LCS_ERR_BAD_ARG;
  *out = nullptr;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMalloc(out, bytes ? bytes : 1));
  return LCS_OK;
}

int lcs_device_free(lcs_ctx *c, void *p) {
  if (!c) return LCS_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  if (p) HIPCHK(c, hipFree(p));
  return LCS_OK;
}

int lcs_device_upload(lcs_ctx *c, void *d_dst, const void *h_src, size_t bytes) {
  if (!c || !d_dst || !h_src) return LCS_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
  return LCS_OK;
}

int lcs_batch_enqueue_host(lcs_ctx *c, const void *h_capbufs, int fmt, int n_buf, uint32_t n_cap, const double *f_search_set,
                           uint16_t n_f, const double *fc_requested, const double *fc_programmed, double fs_programmed,
                           int stage_mask) {
  if (!c) return LCS_ERR_BAD_ARG;
  if (!h_capbufs || n_buf < 1 || (fmt != LCS_FMT_C64 && fmt != LCS_FMT_IQ_U8)) { c->err = "bad argument"; return LCS_ERR_BAD_ARG; }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t bytes = (size_t)n_buf * n_cap * (fmt == LCS_FMT_IQ_U8 ? 2 : sizeof(float2));
  if (bytes > c->h2d_bytes) {
    if (c->st_open) { c->err = "lcs_stream_close first"; return LCS_ERR_BAD_ARG; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->h2d) (void)hipFree(c->h2d);
    c->h2d = nullptr; c->h2d_bytes = 0;
    HIPCHK(c, hipMalloc(&c->h2d, bytes));
    c->h2d_bytes = bytes;
  }
  hipPointerAttribute_t at;
  const bool locked = hipPointerGetAttributes(&at, h_capbufs) == hipSuccess && at.type == hipMemoryTypeHost;
  (void)hipGetLastError();        // an ordinary malloc'ed pointer makes the query fail: not an error here
  if (locked) {
    HIPCHK(c, hipMemcpyAsync(c->h2d, h_capbufs, bytes, hipMemcpyHostToDevice, c->stream));
  } else {
    constexpr size_t CH = (size_t)4 << 20;
    for (int k = 0; k < 2; ++k)
      if (!c->h_stage[k]) {
        HIPCHK(c, hipHostMalloc(&c->h_stage[k], CH, hipHostMallocDefault));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_stage[k], hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(c->ev_stage[k], c->stream));
      }
    int k = 0;
    for (size_t off = 0; off < bytes; off += CH, k ^= 1) {
      const size_t n = std::min(CH, bytes - off);
      HIPCHK(c, hipEventSynchronize(c->ev_stage[k]));          // the DMA that last read this slot is done
      std::memcpy(c->h_stage[k], (const char *)h_capbufs + off, n);
      HIPCHK(c, hipMemcpyAsync((char *)c->h2d + off, c->h_stage[k], n, hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipEventRecord(c->ev_stage[k], c->stream));
    }
  }
  return lcs_batch_enqueue(c, c->h2d, fmt, n_buf, n_cap, f_search_set, n_f, fc_requested, fc_programmed, fs_programmed, stage_mask);
}

int lcs_search_batch_host(lcs_ctx *c, const void *h_capbufs, int fmt, int n_buf, uint32_t n_cap, const double *f_search_set,
                          uint16_t n_f, const double *fc_requested, const double *fc_programmed, double fs_programmed,
                          int stage_mask, lcs_cell *cells, int max_cells_per_buf, int *n_cells) {
  const int rc = lcs_batch_enqueue_host(c, h_capbufs, fmt, n_buf, n_cap, f_search_set, n_f, fc_requested, fc_programmed,
                                        fs_programmed, stage_mask);
  if (rc) return rc;
  return lcs_batch_collect(c, cells, max_cells_per_buf, n_cells);
}

// Debug readback: the xcorr_pss outputs of buffer `buf` of the last batch, in the reference's layouts.
int lcs_batch_readback(lcs_ctx *c, int buf, float *single, double *pow_, int32_t *frq, double *sp_incoherent, double *z_th1) {
  if (!c || c->last_n_buf <= 0 || buf < 0 || buf >= c->last_n_buf) { if (c) c->err = "no such buffer in the last batch"; return LCS_ERR_BAD_ARG; }
  HIPCHK(c, hipSetDevice(c->device));
  const XcGeom &geo = c->last_geo;
  const size_t NE = 3 * LCS_N_IDX;
  int rc;
  if (single) {
    if ((rc = lcs_launch_single_layout(c, geo, buf, c->sref, 1))) return rc;
    HIPCHK(c, hipMemcpyAsync(single, c->sref, sizeof(float) * NE * geo.n_f, hipMemcpyDeviceToHost, c->stream));
  }
  if (pow_) HIPCHK(c, hipMemcpyAsync(pow_, c->pow_ + (size_t)buf * NE, sizeof(double) * NE, hipMemcpyDeviceToHost, c->stream));
  if (frq) HIPCHK(c, hipMemcpyAsync(frq, c->frq + (size_t)buf * NE, sizeof(int) * NE, hipMemcpyDeviceToHost, c->stream));
  if (sp_incoherent) HIPCHK(c, hipMemcpyAsync(sp_incoherent, c->spinc + (size_t)buf * LCS_N_IDX, sizeof(double) * LCS_N_IDX, hipMemcpyDeviceToHost, c->stream));
  if (z_th1) HIPCHK(c, hipMemcpyAsync(z_th1, c->zth + (size_t)buf * LCS_N_IDX, sizeof(double) * LCS_N_IDX, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LCS_OK;
}

// ---------------------------------------------------------- single-cell stage entry points
namespace {
int upload_cap_and_params(lcs_ctx *c, const double *capbuf, uint32_t n_cap, double fc_req, double fc_prog, double fs_prog) {
  int rc;
  if (!capbuf || n_cap < 128) { c->err = "bad capture buffer"; return LCS_ERR_BAD_ARG; }
  HIPCHK(c, hipSetDevice(c->device));
  if ((rc = ensure_ws(c, 1, n_cap, std::max(1, c->cap_n_f), false))) return rc;
  c->h_params = SlotParams{fc_req, fc_prog, fs_prog};      // outlives the asynchronous copy (the callers synchronise later)
  HIPCHK(c, hipMemcpyAsync(c->cap64, capbuf, sizeof(double2) * n_cap, hipMemcpyHostToDevice, c->stream));
  c->cap64_valid = true;
  HIPCHK(c, hipMemcpyAsync(c->params, &c->h_params, sizeof(SlotParams), hipMemcpyHostToDevice, c->stream));
  return LCS_OK;
}
int put_single_work_item(lcs_ctx *c, const lcs_cell *cell, int n_ofdm) {
  int rc;
  if ((rc = ensure_percell(c))) return rc;
  const WorkItem wi{0, 0};
  const int nw[4] = {1, 1, 0, 0};      // one cell taken of one; no re-detections
  const double hdr[3] = {(double)n_ofdm, 0.0, 0.0};
  HIPCHK(c, hipMemcpyAsync(c->work_items, &wi, sizeof(wi), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->n_work, nw, sizeof(nw), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->cells_out, cell, sizeof(lcs_cell), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->cell_scratch, hdr, sizeof(hdr), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));   // the sources above are stack variables
  return LCS_OK;
}
int n_ofdm_for(const lcs_cell *cell) {
  return cell->cp_type == LCS_CP_NORMAL ? 854 : (cell->cp_type == LCS_CP_EXTENDED ? 732 : -1);
}
}  // namespace

int lcs_sss_detect(lcs_ctx *c, const lcs_cell *cell, const double *capbuf, uint32_t n_cap, double thresh2_n_sigma,
                   double fc_req, double fc_prog, double fs_prog, lcs_cell *cell_out, double *h1_np, double *h2_np,
                   double *h1_nrm, double *h2_nrm, double *h1_ext, double *h2_ext, double *ll_nrm, double *ll_ext) {
  if (!c || !cell || !cell_out) return LCS_ERR_BAD_ARG;
  if (cell->n_id_2 < 0 || cell->n_id_2 > 2) { c->err = "cell.n_id_2 must be 0..2"; return LCS_ERR_BAD_ARG; }
  int rc;
  if ((rc = upload_cap_and_params(c, capbuf, n_cap, fc_req, fc_prog, fs_prog))) return rc;
  if ((rc = ensure_percell(c))) return rc;
  const int one = 1;
  HIPCHK(c, hipMemcpyAsync(c->peaks, cell, sizeof(lcs_cell), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->npeaks, &one, sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(c->d_dbg, 0, sizeof(double) * 2048, c->stream));
  if ((rc = lcs_launch_sss_only(c, n_cap, thresh2_n_sigma, c->d_dbg))) return rc;
  std::vector<double> dbg(1292);
  HIPCHK(c, hipMemcpyAsync(cell_out, c->peaks, sizeof(lcs_cell), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(dbg.data(), c->d_dbg, sizeof(double) * dbg.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (h1_np) std::memcpy(h1_np, &dbg[0], 62 * sizeof(double));
  if (h2_np) std::memcpy(h2_np, &dbg[62], 62 * sizeof(double));
  if (h1_nrm) std::memcpy(h1_nrm, &dbg[124], 124 * sizeof(double));
  if (h2_nrm) std::memcpy(h2_nrm, &dbg[248], 124 * sizeof(double));
  if (h1_ext) std::memcpy(h1_ext, &dbg[372], 124 * sizeof(double));
  if (h2_ext) std::memcpy(h2_ext, &dbg[496], 124 * sizeof(double));
  if (ll_nrm) std::memcpy(ll_nrm, &dbg[620], 336 * sizeof(double));
  if (ll_ext) std::memcpy(ll_ext, &dbg[956], 336 * sizeof(double));
  return LCS_OK;
}

int lcs_pss_sss_foe(lcs_ctx *c, const lcs_cell *cell_in, const double *capbuf, uint32_t n_cap, double fc_req,
                    double fc_prog, double fs_prog, lcs_cell *cell_out) {
  if (!c || !cell_in || !cell_out) return LCS_ERR_BAD_ARG;
  if (n_ofdm_for(cell_in) < 0 || cell_in->n_id_1 < 0 || cell_in->n_id_1 > 167 || cell_in->n_id_2 < 0 || cell_in->n_id_2 > 2) {
    c->err = "pss_sss_foe needs a cell with n_id_1, n_id_2 and a known cp_type";   // the reference throws (src/searcher.cpp:786)
    return LCS_ERR_BAD_ARG;
  }
  int rc;
  if ((rc = upload_cap_and_params(c, capbuf, n_cap, fc_req, fc_prog, fs_prog))) return rc;
  const int one = 1;
  HIPCHK(c, hipMemcpyAsync(c->peaks, cell_in, sizeof(lcs_cell), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->npeaks, &one, sizeof(int), hipMemcpyHostToDevice, c->stream));
  if ((rc = lcs_launch_foe_only(c, n_cap))) return rc;
  HIPCHK(c, hipMemcpyAsync(cell_out, c->peaks, sizeof(lcs_cell), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LCS_OK;
}

int lcs_extract_tfg(lcs_ctx *c, const lcs_cell *cell, const double *capbuf, uint32_t n_cap, double fc_req,
                    double fc_prog, double fs_prog, double *tfg, double *tfg_timestamp, int *n_ofdm) {
  if (!c || !cell || !tfg || !tfg_timestamp || !n_ofdm) return LCS_ERR_BAD_ARG;
  const int no = n_ofdm_for(cell);
  if (no < 0) { c->err = "extract_tfg needs a known cp_type"; return LCS_ERR_BAD_ARG; }   // ref :883 throws
  int rc;
  if ((rc = upload_cap_and_params(c, capbuf, n_cap, fc_req, fc_prog, fs_prog))) return rc;
  if ((rc = put_single_work_item(c, cell, no))) return rc;
  c->needed_rows_only = false;
  if ((rc = lcs_launch_tfg(c, n_cap, false))) return rc;
  double oob = 0;
  HIPCHK(c, hipMemcpyAsync(tfg, c->tfg, sizeof(double2) * no * LCS_TFG_NSC, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(tfg_timestamp, c->tfg_ts, sizeof(double) * no, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&oob, c->cell_scratch + 2, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *n_ofdm = no;
  if (oob != 0.0) { c->err = "a DFT window falls outside the capture buffer (the reference would read out of bounds)"; return LCS_ERR_BAD_ARG; }
  return LCS_OK;
}

int lcs_tfoec(lcs_ctx *c, const lcs_cell *cell, const double *tfg, const double *tfg_timestamp, int n_ofdm,
              double fc_req, double fc_prog, double *tfg_comp, double *tfg_comp_timestamp, lcs_cell *cell_out) {
  if (!c || !cell || !tfg || !tfg_timestamp || !tfg_comp || !tfg_comp_timestamp || !cell_out) return LCS_ERR_BAD_ARG;
  if (n_ofdm_for(cell) < 0 || n_ofdm < 14 || n_ofdm > LCS_TFG_MAX_OFDM || cell->n_id_1 < 0 || cell->n_id_2 < 0) {
    c->err = "tfoec needs a detected cell and 14..854 OFDM symbols";
    return LCS_ERR_BAD_ARG;
  }
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  if ((rc = ensure_ws(c, 1, std::max<uint32_t>(c->cap_n_cap, 153600), std::max(1, c->cap_n_f), false))) return rc;
  SlotParams p{fc_req, fc_prog, 0.0};
  HIPCHK(c, hipMemcpyAsync(c->params, &p, sizeof(p), hipMemcpyHostToDevice, c->stream));
  if ((rc = put_single_work_item(c, cell, n_ofdm))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->tfg, tfg, sizeof(double2) * n_ofdm * LCS_TFG_NSC, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->tfg_ts, tfg_timestamp, sizeof(double) * n_ofdm, hipMemcpyHostToDevice, c->stream));
  if ((rc = lcs_launch_rs_build(c))) return rc;
  c->needed_rows_only = false;
  if ((rc = lcs_launch_tfoec(c, true))) return rc;
  HIPCHK(c, hipMemcpyAsync(tfg_comp, c->tfg_comp, sizeof(double2) * n_ofdm * LCS_TFG_NSC, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(tfg_comp_timestamp, c->tfg_ts_comp, sizeof(double) * n_ofdm, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(cell_out, c->cells_out, sizeof(lcs_cell), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LCS_OK;
}

int lcs_decode_mib(lcs_ctx *c, const lcs_cell *cell, const double *tfg, int n_ofdm, lcs_cell *cell_out) {
  if (!c || !cell || !tfg || !cell_out) return LCS_ERR_BAD_ARG;
  const int need = n_ofdm_for(cell);
  if (need < 0 || n_ofdm != need || cell->n_id_1 < 0 || cell->n_id_2 < 0) {
    c->err = "decode_mib needs a detected cell and its full 854/732-symbol grid";
    return LCS_ERR_BAD_ARG;
  }
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  if ((rc = ensure_ws(c, 1, std::max<uint32_t>(c->cap_n_cap, 153600), std::max(1, c->cap_n_f), false))) return rc;      // (a fresh context: the kernels read the slot's parameter record)
  if ((rc = put_single_work_item(c, cell, n_ofdm))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->tfg_comp, tfg, sizeof(double2) * n_ofdm * LCS_TFG_NSC, hipMemcpyHostToDevice, c->stream));
  if ((rc = lcs_launch_rs_build(c))) return rc;
  c->needed_rows_only = true;      // decode_mib reads the channel estimate on PBCH rows only
  if ((rc = lcs_launch_mib(c, false))) return rc;
  HIPCHK(c, hipMemcpyAsync(cell_out, c->cells_out, sizeof(lcs_cell), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LCS_OK;
}

// chan_est of decode_mib as a stage of its own (ref src/searcher.cpp:1369-1477, ce_interp_hex :1223-1362): the channel
// estimate of one antenna port on the whole grid and its noise power -- an internal function of the reference, exported
// so that the tests can pin it to the oracle directly rather than through the decoded MIB.
int lcs_chan_est(lcs_ctx *c, const lcs_cell *cell, const double *tfg, int n_ofdm, int port, double *ce_tfg, double *np_out) {
  if (!c || !cell || !tfg || !ce_tfg || !np_out) return LCS_ERR_BAD_ARG;
  const int need = n_ofdm_for(cell);
  if (need < 0 || n_ofdm != need || cell->n_id_1 < 0 || cell->n_id_2 < 0 || port < 0 || port > 3) {
    c->err = "chan_est needs a detected cell, its full 854/732-symbol grid and a port 0..3";
    return LCS_ERR_BAD_ARG;
  }
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  if ((rc = ensure_ws(c, 1, std::max<uint32_t>(c->cap_n_cap, 153600), std::max(1, c->cap_n_f), false))) return rc;      // (a fresh context: the kernels read the slot's parameter record)
  if ((rc = put_single_work_item(c, cell, n_ofdm))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->tfg_comp, tfg, sizeof(double2) * n_ofdm * LCS_TFG_NSC, hipMemcpyHostToDevice, c->stream));
  if ((rc = lcs_launch_rs_build(c))) return rc;
  c->needed_rows_only = false;
  if ((rc = lcs_launch_chan_est(c))) return rc;
  int first, per_port, n_rs_at;
  lcs_chan_est_np_layout(&first, &per_port, &n_rs_at);
  std::vector<double> sc(LCS_CELL_SCRATCH);
  HIPCHK(c, hipMemcpyAsync(ce_tfg, c->ce + (size_t)port * LCS_TFG_ROWS * LCS_TFG_NSC, sizeof(double2) * n_ofdm * LCS_TFG_NSC, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(sc.data(), c->cell_scratch, sizeof(double) * LCS_CELL_SCRATCH, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  double s = 0;
  for (int q = 0; q < per_port; ++q) s += sc[first + port * per_port + q];      // chunk partials, in chunk order (tfg_mib.hip np_from_partials)
  *np_out = s / (sc[n_rs_at + port] * 12);
  return LCS_OK;
}

// One host buffer through the whole chain (ref src/CellSearch.cpp:484-558).
int lcs_search_capbuf(lcs_ctx *c, const double *capbuf, uint32_t n_cap, const double *f_search_set, uint16_t n_f,
                      double fc_req, double fc_prog, double fs_prog, lcs_cell *cells, int max_cells, int *n_cells,
                      lcs_cell *peaks, int max_peaks, int *n_peaks) {
  int rc = check_common(c, n_cap, n_f);
  if (rc) return rc;
  if (!capbuf || !f_search_set || !n_cells || (max_cells > 0 && !cells)) { c->err = "null argument"; return LCS_ERR_BAD_ARG; }
  HIPCHK(c, hipSetDevice(c->device));
  XcGeom geo;
  if ((rc = upload_host_capbuf(c, capbuf, n_cap, f_search_set, n_f, 2, fc_req, fc_prog, fs_prog, false, &geo))) return rc;
  if ((rc = ensure_percell(c))) return rc;
  c->repair_peaks_only = true;      // no array leaves this call: the peak list is what has to be exact
  rc = lcs_launch_xcorr(c, 1, geo, false, false);
  c->repair_peaks_only = false;
  if (rc) return rc;
  if ((rc = lcs_launch_peak_search(c, 1, geo, std::pow(10.0, -12.0 / 10.0), true))) return rc;
  if ((rc = lcs_launch_sss_foe(c, 1, n_cap, 3.0, nullptr))) return rc;
  c->needed_rows_only = true;
  if ((rc = lcs_launch_gather_work(c, 1, 0))) return rc;
  if ((rc = lcs_launch_tfg(c, n_cap, true))) return rc;
  if ((rc = lcs_launch_tfoec(c, false))) return rc;
  if ((rc = lcs_launch_mib(c, true))) return rc;
  std::vector<lcs_cell> tmp(LCS_MAXP);
  int np = 0;
  HIPCHK(c, hipMemcpyAsync(tmp.data(), c->peaks, sizeof(lcs_cell) * LCS_MAXP, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&np, c->npeaks, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  rc = LCS_OK;
  if (np > LCS_MAXP) { np = LCS_MAXP; rc = LCS_ERR_OVERFLOW; }
  int n = 0;
  for (int i = 0; i < np; ++i) {
    if (peaks && i < max_peaks) {   // the peak_search view of the record (PSS fields only)
      lcs_cell pk;
      lcs_cell_init(&pk);
      pk.fc_requested = tmp[i].fc_requested; pk.fc_programmed = tmp[i].fc_programmed; pk.pss_pow = tmp[i].pss_pow;
      pk.ind = tmp[i].ind; pk.freq = tmp[i].freq; pk.n_id_2 = tmp[i].n_id_2;
      peaks[i] = pk;
    }
    if (tmp[i].n_id_1 == -1 || tmp[i].n_rb_dl == -1) continue;
    if (n < max_cells) cells[n] = tmp[i]; else rc = LCS_ERR_OVERFLOW;
    ++n;
  }
  if (n_peaks) *n_peaks = np;
  *n_cells = n;
  if (rc) c->err = "more results than the output arrays hold";
  return rc;
}

// ------------------------------------------------------------------- one buffer, hypotheses split over GPUs
// SURVEY 8e "latency mode": every rank correlates its contiguous share of f_search_set; the shares meet where the
// reference takes the maximum over the frequency axis (src/searcher.cpp:369-382).  RCCL has no arg-max: the collapsed
// power (a non-negative float: ordered like its bit pattern) and the complemented GLOBAL hypothesis index are packed
// into one 64-bit word, (bits(pow) << 32) | (0xFFFFFFFF - foi), and an ordinary MAX all-reduce then reproduces the
// reference's first-maximum rule (strict >: the lowest index wins a tie, :374).  The words never leave the GPUs:
// lcs_foe_partial leaves them in a device buffer of the caller (a torch tensor handed to torch.distributed), the caller
// all-reduces in place, lcs_foe_finish reads them back in, runs peak_search (identical on every rank) and the per-peak
// stages, and reports the cells whose winning hypothesis this rank owns (it alone holds the xc_incoherent_single slice
// the refinement of `ind` reads, :457-465).
int lcs_foe_partial(lcs_ctx *c, const double *capbuf, uint32_t n_cap, const double *f_search_set, uint16_t n_f, int f_first, int f_count,
                    double fc_req, double fc_prog, double fs_prog, void *d_words, double *d_meta) {
  int rc = check_common(c, n_cap, n_f);
  if (rc) return rc;
  if (!capbuf || !f_search_set || !d_words || !d_meta || f_first < 0 || f_count < 0 || f_first + f_count > n_f) { c->err = "bad argument"; return LCS_ERR_BAD_ARG; }
  HIPCHK(c, hipSetDevice(c->device));
  c->foe_ready = false;
  // a rank without hypotheses still takes part: it correlates one (the first) so that its buffer, tables and power
  // estimate exist, and contributes words that never win
  const int cnt = std::max(1, f_count), first = f_count ? f_first : 0;
  XcGeom geo;
  if ((rc = ensure_ws(c, 1, n_cap, n_f, false))) return rc;      // lcs_foe_finish puts the WHOLE grid into fset (this rank correlates its share only)
  if ((rc = upload_host_capbuf(c, capbuf, n_cap, f_search_set + first, cnt, 2, fc_req, fc_prog, fs_prog, false, &geo))) return rc;
  if ((rc = ensure_percell(c))) return rc;
  // no tie repair here: a near-tie may span two ranks' shares -- lcs_foe_contend settles them after the all-reduce, identically
  // for every split; the collapse keeps the runner-up values for it
  c->skip_frq_repair = true;
  rc = lcs_launch_xcorr(c, 1, geo, false, false);
  c->skip_frq_repair = false;
  if (rc) return rc;
  geo.foi0 = f_count ? f_first : -1;            // -1: owns nothing
  if ((rc = lcs_launch_foe_pack(c, geo, static_cast<long long *>(d_words), d_meta))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));    // the caller's collective runs on another stream
  c->foe_geo = geo;
  c->foe_n_cap = n_cap;
  c->foe_ready = true;
  return LCS_OK;
}

// Exactness of the split (round 5).  The packed MAX decides between float values that the kernels reproduce to ~1e-7: where two
// hypotheses -- of one rank or of two -- lie closer than LCS_FRQ_TIE_EPS, the winner has to be decided in the reference's own
// arithmetic (k_frq_repair).  After the all-reduce of d_words every rank knows the global maximum of every position; it CONTENDS
// for a position when a hypothesis of its own other than the winner lies within the distance of that maximum, recomputes its
// contenders AND the global winner exactly (it holds the whole buffer and the whole grid), and writes the packed exact first
// maximum into d_words2 (-1 elsewhere).  The caller MAX-all-reduces d_words2 -- the exact first maximum over every contender of
// every rank, identical whatever the split -- and lcs_foe_resolve puts those words in place of the approximate ones.
int lcs_foe_contend(lcs_ctx *c, const double *f_search_set, uint16_t n_f, const void *d_words, void *d_words2) {
  if (!c) return LCS_ERR_BAD_ARG;
  if (!c->foe_ready) { c->err = "lcs_foe_contend needs the lcs_foe_partial call of the same buffer first"; return LCS_ERR_BAD_ARG; }
  if (!f_search_set || !d_words || !d_words2 || n_f < 1 || n_f > LCS_NF_LIMIT) { c->err = "bad argument"; return LCS_ERR_BAD_ARG; }
  HIPCHK(c, hipSetDevice(c->device));
  if (n_f > c->fset_g_cap) {      // the whole grid (fset holds this rank's share)
    int rc_;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if ((rc_ = dev_alloc(c, &c->fset_g, (size_t)n_f))) return rc_;
    c->fset_g_cap = n_f;
  }
  HIPCHK(c, hipMemcpyAsync(c->fset_g, f_search_set, sizeof(double) * n_f, hipMemcpyHostToDevice, c->stream));
  int rc;
  if ((rc = lcs_launch_foe_contend(c, c->foe_geo, c->fset_g, static_cast<const long long *>(d_words), static_cast<long long *>(d_words2)))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));      // the caller's collective runs on another stream (and f_search_set may go away)
  return LCS_OK;
}

int lcs_foe_resolve(lcs_ctx *c, void *d_words, const void *d_words2) {
  if (!c || !d_words || !d_words2) return LCS_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  if ((rc = lcs_launch_foe_resolve(c, static_cast<long long *>(d_words), static_cast<const long long *>(d_words2)))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LCS_OK;
}

int lcs_foe_finish(lcs_ctx *c, const void *d_words, const double *d_meta, const double *f_search_set, uint16_t n_f, lcs_cell *cells,
                   int32_t *order, int max_cells, int *n_cells, lcs_cell *peaks, int max_peaks, int *n_peaks) {
  if (!c) return LCS_ERR_BAD_ARG;
  if (!c->foe_ready) { c->err = "lcs_foe_finish needs the lcs_foe_partial call of the same buffer first"; return LCS_ERR_BAD_ARG; }
  if (!d_words || !d_meta || !f_search_set || !n_cells || (max_cells > 0 && (!cells || !order)) || n_f < 1 || n_f > LCS_NF_LIMIT) { c->err = "bad argument"; return LCS_ERR_BAD_ARG; }
  HIPCHK(c, hipSetDevice(c->device));
  c->foe_ready = false;
  const XcGeom geo = c->foe_geo;
  const uint32_t n_cap = c->foe_n_cap;
  int rc;
  // peak_search names the winning hypothesis by its GLOBAL index: the whole grid now, not this rank's share
  HIPCHK(c, hipMemcpyAsync(c->fset, f_search_set, sizeof(double) * n_f, hipMemcpyHostToDevice, c->stream));
  if ((rc = lcs_launch_foe_unpack(c, geo, static_cast<const long long *>(d_words), d_meta))) return rc;
  if ((rc = lcs_launch_peak_search(c, 1, geo, std::pow(10.0, -12.0 / 10.0), true))) return rc;
  if ((rc = lcs_launch_sss_foe(c, 1, n_cap, 3.0, nullptr))) return rc;
  c->needed_rows_only = true;
  if ((rc = lcs_launch_gather_work(c, 1, 0))) return rc;
  if ((rc = lcs_launch_tfg(c, n_cap, true))) return rc;
  if ((rc = lcs_launch_tfoec(c, false))) return rc;
  if ((rc = lcs_launch_mib(c, true))) return rc;
  std::vector<lcs_cell> tmp(LCS_MAXP);
  int np = 0;
  HIPCHK(c, hipMemcpyAsync(tmp.data(), c->peaks, sizeof(lcs_cell) * LCS_MAXP, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&np, c->npeaks, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  rc = LCS_OK;
  if (np > LCS_MAXP) { np = LCS_MAXP; rc = LCS_ERR_OVERFLOW; }
  int n = 0;
  for (int i = 0; i < np; ++i) {
    const bool mine = tmp[i].reserved == 0;                   // the fused peak search marks peaks won by another rank's hypothesis
    if (peaks && i < max_peaks) {
      lcs_cell pk;
      lcs_cell_init(&pk);
      pk.fc_requested = tmp[i].fc_requested; pk.fc_programmed = tmp[i].fc_programmed; pk.pss_pow = tmp[i].pss_pow;
      pk.ind = mine ? tmp[i].ind : -1; pk.freq = tmp[i].freq; pk.n_id_2 = tmp[i].n_id_2; pk.reserved = tmp[i].reserved;
      peaks[i] = pk;
    }
    if (!mine || tmp[i].n_id_1 == -1 || tmp[i].n_rb_dl == -1) continue;
    if (n < max_cells) { cells[n] = tmp[i]; order[n] = i; } else rc = LCS_ERR_OVERFLOW;
    ++n;
  }
  if (n_peaks) *n_peaks = np;
  *n_cells = n;
  if (rc) c->err = "more results than the output arrays hold";
  return rc;
}

// ---------------------------------------------------------------------------- streaming mode
// LTE-Tracker's searcher thread (ref src/searcher_thread.cpp:83-246) runs the whole chain on every
// 80 ms capture buffer with a single frequency hypothesis (the tracked frequency offset, :97-98) and
// skips cells that are already tracked (:157-177).  Per buffer that is ~25 small launches: here the
// chain is captured ONCE as a hipGraph and replayed per push.  Everything that changes between
// pushes (samples, frequency offset, tracked identities) travels through fixed pinned host buffers
// that the graph's copy nodes read at execution time.
namespace {
// the chain as slot k sees it: its own pinned input buffer and parameter / result block, the shared device workspace
int stream_chain_launches(lcs_ctx *c, int k);
int stream_chain(lcs_ctx *c, int k) {
  // the chain's kernels take the slot parameters and the hypothesis from the stream's device mirror (one copy per push), not from
  // the workspace arrays the other entry points fill: the context's pointers are swapped while the launches are issued / recorded
  SlotParams *params = c->params;
  double *fset = c->fset;
  c->params = reinterpret_cast<SlotParams *>(c->st_dmirror + offsetof(StreamHost, p));
  c->fset = reinterpret_cast<double *>(c->st_dmirror + offsetof(StreamHost, f));
  const int rc = stream_chain_launches(c, k);
  c->params = params;
  c->fset = fset;
  return rc;
}
int stream_chain_launches(lcs_ctx *c, int k) {
  StreamHost *h = c->st_host[k];
  const XcGeom geo = make_geo(c->st_n_cap, 1, 2);
  int rc;
  c->use_i8 = c->st_fmt == LCS_FMT_IQ_U8;      // one hypothesis: no window-start spread, the int8 kernel always fits
  c->use_f16 = false;
  c->foe_ready = false;
  HIPCHK(c, hipMemcpyAsync(c->st_din, c->st_hin[k], c->st_in_bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->st_dmirror, h, LCS_STREAM_IN_BYTES, hipMemcpyHostToDevice, c->stream));      // parameters, tracked list, hypothesis: one copy
  if ((rc = lcs_launch_ingest(c, c->st_din, c->st_fmt, 1, c->st_n_cap))) return rc;
  if ((rc = lcs_launch_xcorr(c, 1, geo, false, false))) return rc;
  if ((rc = lcs_launch_peak_search(c, 1, geo, std::pow(10.0, -12.0 / 10.0), true))) return rc;
  if ((rc = lcs_launch_sss_foe(c, 1, c->st_n_cap, 3.0, nullptr))) return rc;
  c->needed_rows_only = true;
  if ((rc = lcs_launch_gather_work(c, 1, 0))) return rc;
  if ((rc = lcs_launch_tfg(c, c->st_n_cap, true))) return rc;
  if ((rc = lcs_launch_tfoec(c, false))) return rc;
  if ((rc = lcs_launch_mib(c, true))) return rc;
  HIPCHK(c, hipMemcpyAsync(h->res, c->peaks, sizeof(h->res), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&h->n_peaks, c->npeaks, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(h->n_work, c->n_work, sizeof(h->n_work), hipMemcpyDeviceToHost, c->stream));
  return LCS_OK;
}
}  // namespace

int lcs_stream_close(lcs_ctx *c) {
  if (!c) return LCS_ERR_BAD_ARG;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (int k = 0; k < 2; ++k) {
    if (c->st_exec[k]) (void)hipGraphExecDestroy(c->st_exec[k]);
    if (c->st_graph[k]) (void)hipGraphDestroy(c->st_graph[k]);
    if (c->st_hin[k]) (void)hipHostFree(c->st_hin[k]);
    if (c->st_host[k]) (void)hipHostFree(c->st_host[k]);
    if (c->st_ev0[k]) (void)hipEventDestroy(c->st_ev0[k]);
    if (c->st_ev1[k]) (void)hipEventDestroy(c->st_ev1[k]);
    c->st_exec[k] = nullptr; c->st_graph[k] = nullptr; c->st_hin[k] = nullptr; c->st_host[k] = nullptr; c->st_ev0[k] = c->st_ev1[k] = nullptr;
  }
  if (c->st_din) (void)hipFree(c->st_din);
  if (c->st_dmirror) (void)hipFree(c->st_dmirror);
  c->st_din = nullptr; c->st_dtracked = nullptr; c->st_dntracked = nullptr; c->st_dmirror = nullptr;
  c->st_open = false;
  c->st_head = c->st_count = 0;
  c->single_stream = false;
  return LCS_OK;
}

// Two slots: the chain is captured twice, once per pinned input buffer + parameter / result block, so that the host may
// fill and launch buffer i + 1 while the graph of buffer i is still running (the launches themselves serialise on the
// context's stream and share the device workspace).
int lcs_stream_open(lcs_ctx *c, int fmt, uint32_t n_cap, double fc_requested, double fc_programmed, double fs_programmed) {
  int rc = check_common(c, n_cap, 1);
  if (rc) return rc;
  if (fmt != LCS_FMT_C64 && fmt != LCS_FMT_IQ_U8) { c->err = "unknown capture format"; return LCS_ERR_BAD_ARG; }
  if (c->st_open) lcs_stream_close(c);
  HIPCHK(c, hipSetDevice(c->device));
  if ((rc = ensure_ws(c, 1, n_cap, 1, false))) return rc;
  if ((rc = ensure_percell(c))) return rc;
  if (fmt == LCS_FMT_IQ_U8 && (rc = ensure_i8(c))) return rc;
  if ((rc = lcs_ensure_btab(c))) return rc;      // host buffers that are not dongle data take the fp32 kernel, also while the stream is open
  c->st_fmt = fmt;
  c->st_n_cap = n_cap;
  c->st_in_bytes = (size_t)n_cap * (fmt == LCS_FMT_IQ_U8 ? 2 : sizeof(float2));
  HIPCHK(c, hipMalloc(&c->st_din, c->st_in_bytes));
  HIPCHK(c, hipMalloc((void **)&c->st_dmirror, LCS_STREAM_IN_BYTES));
  c->st_dntracked = reinterpret_cast<int *>(c->st_dmirror + offsetof(StreamHost, n_tracked));
  c->st_dtracked = reinterpret_cast<int16_t *>(c->st_dmirror + offsetof(StreamHost, tracked));
  for (int k = 0; k < 2; ++k) {
    HIPCHK(c, hipHostMalloc(&c->st_hin[k], c->st_in_bytes, hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc((void **)&c->st_host[k], sizeof(StreamHost), hipHostMallocDefault));
    HIPCHK(c, hipEventCreate(&c->st_ev0[k]));
    HIPCHK(c, hipEventCreate(&c->st_ev1[k]));
    std::memset(c->st_hin[k], fmt == LCS_FMT_IQ_U8 ? 127 : 0, c->st_in_bytes);
    std::memset(c->st_host[k], 0, sizeof(StreamHost));
    c->st_host[k]->p = SlotParams{fc_requested, fc_programmed, fs_programmed};
  }
  c->cap64_valid = false;
  c->single_stream = true;
  c->st_open = true;
  c->st_head = c->st_count = 0;
  // one eager pass (lazy allocations, function attributes), then the same call sequence under capture, per slot
  if ((rc = stream_chain(c, 0))) { lcs_stream_close(c); return rc; }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int k = 0; k < 2; ++k) {
    HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    rc = stream_chain(c, k);
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(c->stream, &g);
    if (rc || e != hipSuccess || !g) {
      if (!rc) { c->err = std::string("hipStreamEndCapture: ") + hipGetErrorString(e); rc = LCS_ERR_HIP; }
      lcs_stream_close(c);
      return rc;
    }
    c->st_graph[k] = g;
    HIPCHK(c, hipGraphInstantiate(&c->st_exec[k], c->st_graph[k], nullptr, nullptr, 0));
  }
  return LCS_OK;
}

int lcs_stream_push(lcs_ctx *c, const void *samples, double f_off, const int16_t *tracked_ids, int n_tracked) {
  if (!c || !c->st_open || !samples) { if (c) c->err = "stream not open"; return LCS_ERR_BAD_ARG; }
  if (c->st_count >= 2) { c->err = "two buffers are in flight already: lcs_stream_collect first"; return LCS_ERR_BAD_ARG; }
  if (n_tracked < 0 || n_tracked > 504 || (n_tracked > 0 && !tracked_ids)) { c->err = "bad tracked list"; return LCS_ERR_BAD_ARG; }
  HIPCHK(c, hipSetDevice(c->device));
  c->foe_ready = false;      // the graph overwrites slot 0
  const int k = (c->st_head + c->st_count) & 1;
  std::memcpy(c->st_hin[k], samples, c->st_in_bytes);
  c->st_host[k]->f = f_off;
  c->st_host[k]->n_tracked = n_tracked;
  for (int i = 0; i < n_tracked; ++i) c->st_host[k]->tracked[i] = tracked_ids[i];
  HIPCHK(c, hipEventRecord(c->st_ev0[k], c->stream));
  HIPCHK(c, hipGraphLaunch(c->st_exec[k], c->stream));
  HIPCHK(c, hipEventRecord(c->st_ev1[k], c->stream));
  ++c->st_count;
  return LCS_OK;
}

// Results of the OLDEST buffer in flight.
int lcs_stream_collect(lcs_ctx *c, lcs_cell *cells, int max_cells, int *n_cells, int *n_redetected, float *gpu_ms) {
  if (!c || !c->st_open || c->st_count < 1 || !n_cells || (max_cells > 0 && !cells)) { if (c) c->err = "nothing to collect"; return LCS_ERR_BAD_ARG; }
  HIPCHK(c, hipSetDevice(c->device));
  const int k = c->st_head;
  HIPCHK(c, hipEventSynchronize(c->st_ev1[k]));
  c->st_head ^= 1;
  --c->st_count;
  if (gpu_ms) HIPCHK(c, hipEventElapsedTime(gpu_ms, c->st_ev0[k], c->st_ev1[k]));
  const StreamHost *h = c->st_host[k];
  int rc = LCS_OK, n = 0;
  const int np = std::min(h->n_peaks, (int)LCS_MAXP);
  if (h->n_peaks > LCS_MAXP || h->n_work[1] > std::min(c->max_work, c->percell_cap)) rc = LCS_ERR_OVERFLOW;
  // The reference appends a decoded cell to the tracked list AT ONCE (ref src/searcher_thread.cpp:233-236), so a later peak of the
  // same buffer that sss_detect gives the same identity -- a second path of a fading channel, a sidelobe -- meets "already being
  // tracked" (:157-177) before anything else is done with it.  The chain here decodes every peak in parallel; in peak order the
  // first decoded record of an identity is the one the reference keeps, the later peaks of that identity count as seen again.
  int seen[LCS_MAXP], n_seen = 0, again = 0;
  for (int i = 0; i < np; ++i) {
    const lcs_cell &pc = h->res[i];
    if (pc.n_id_1 == -1) continue;                          // no SSS
    const int id = pc.n_id_2 + 3 * pc.n_id_1;
    bool known = false;
    for (int q = 0; q < n_seen; ++q) known = known || seen[q] == id;
    if (known) { ++again; continue; }
    if (pc.n_rb_dl == -1) continue;                         // no MIB / tracked before this buffer (never decoded: counted on the device)
    if (n < max_cells) cells[n] = pc; else rc = LCS_ERR_OVERFLOW;
    ++n;
    seen[n_seen++] = id;
  }
  *n_cells = n;
  if (n_redetected) *n_redetected = h->n_work[2] + again;
  if (rc) c->err = "more results than the output array holds";
  return rc;
}

int lcs_last_xcorr_ms(lcs_ctx *c, float *ms, int *n_launches) {
  if (!c || !ms) return LCS_ERR_BAD_ARG;
  HIPCHK(c, hipEventSynchronize(c->ev_xc1));
  HIPCHK(c, hipEventElapsedTime(ms, c->ev_xc0, c->ev_xc1));
  if (n_launches) *n_launches = c->last_xc_launches;
  return LCS_OK;
}

int lcs_last_frq_repairs(lcs_ctx *c, int *n_positions) {
  if (!c || !n_positions || !c->n_fix) return LCS_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(n_positions, c->n_fix, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LCS_OK;
}

int lcs_last_frq_repair_stats(lcs_ctx *c, int *n_listed, int *n_unrepaired) {
  if (!c || !n_listed || !n_unrepaired || !c->n_fix) return LCS_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  int v[2] = {0, 0};
  HIPCHK(c, hipMemcpyAsync(v, c->n_fix, sizeof(v), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *n_listed = v[0];
  *n_unrepaired = v[1];
  return LCS_OK;
}

int lcs_last_xcorr_info(lcs_ctx *c, double *executed_ops, const char **kernel) {
  if (!c) return LCS_ERR_BAD_ARG;
  if (executed_ops) *executed_ops = c->last_xc_ops;
  if (kernel) *kernel = c->last_xc_kernel;
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
