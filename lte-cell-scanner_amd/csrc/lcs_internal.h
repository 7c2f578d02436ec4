// lcs_internal.h -- shared declarations of the MI355X-native searcher (not installed).
//
// Device data layout (everything lives in one workspace, slot-major; a "slot" is one
// capture buffer in flight):
//   cap32   [S][n_cap]            float2   capture buffer, fp32 (PSS correlation input)
//   cap64   [1][n_cap]            double2  fp64 copy, only when a host entry point hands over complex<double>
//   tmpl    [S][n_f][3][137]      float2   conj(fshift(pss_td))/137   (searcher.cpp:146-151)
//   start   [S][NW][n_f]          int      round_i(m*.005*k_factor*fs) (searcher.cpp:298)
//   smin/kp2[S][NW][G]            int      per (window, 16-template group): first lag offset, tap pairs
//   btab    [S][NW][G][KP2][64]   float    MFMA B operands: delay-shifted templates (fp32 kernel)
//   brow8   [S][G][LCS_I8_IMG]    uint32   int8 kernel: three-digit operand rows of a group, as they sit in LDS
//   single  [S][G][9600][16]      float    xc_incoherent_single, group-major (16 templates = one 64 B row)
//   sref    [3][9600][n_f]        float    reference-layout staging of single for the stage entry points
//   pow/frq [S][3][9600]          double/int
//   spinc/zth [S][9600]           double
//   peaks   [S][MAXP] lcs_cell, npeaks [S]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstddef>
#include <string>
#include <vector>
#include "../../include/lcs.h"

#define LCS_NW_MAX 16        // incoherent-combining windows (15 for a 153600-sample buffer)
// Frequency hypotheses per call: the reference loops over whatever f_search_set holds (src/searcher.cpp:113-174; CellSearch builds
// n_f = 2 floor((fc ppm / 1e6 + 2500) / 5000) + 1, src/CellSearch.cpp:463-465: 125 at 2.6 GHz, 289 at 6 GHz for the default 120 ppm).
// Every table is sized per call (lcs_api.hip: ensure_ws) and laid out with the call's own strides; this is a sanity bound only
// (+-2.56 MHz of search span, ~120 MB of xc_incoherent_single per buffer).
#define LCS_NF_LIMIT 1024
// Every kernel other than the PSS correlation is small and latency-bound; in the pipelined chain it
// shares CUs with the next batch's correlation waves.  Raising the wave priority lets the SIMD
// arbiter issue these few waves ahead of the MFMA stream instead of round-robin behind 4-5 of them.
#define LCS_TAIL_PRIO() __builtin_amdgcn_s_setprio(3)
#define LCS_TG 16            // templates per MFMA column group
#define LCS_KP2_MAX 128      // tap pairs per (window, group): 137 taps + up to 119 samples of spread
#define LCS_KP2_UNROLL 4
#define LCS_LAG_TILE 64      // lags per wave
#define LCS_PS 336           // LDS plane stride in floats: >= 64 + 2*KP2_MAX, == 16 (mod 32)
#define LCS_MAXP LCS_MAX_PEAKS   // peaks kept per capture buffer: the most peak_search can return (lcs.h)
// xc_incoherent_collapsed_frq: positions whose two best hypotheses lie within this (relative) of each other are recomputed in the
// reference's arithmetic (k_frq_repair, pss_xcorr.hip).  The correlation kernels' values deviate from the reference's by ~1e-7
// (worst element ever measured: 1e-6 of the buffer's largest); two values further apart than this keep their order.
#define LCS_FRQ_TIE_EPS 4e-6f
#define LCS_I8_KB 5          // 32-tap blocks of the int8 correlation kernel
#define LCS_I8_OFF 16        // int8 kernel: a template column's delay inside its group (window-start spread) stays below this
#define LCS_I8_MAX_TAPS (137 + LCS_I8_OFF - 1)
#define LCS_I8_IMG 17024     // dwords of the int8 kernel's operand image per (buffer, group) (pss_xcorr_i8.hip)
#define LCS_F16_IMG 11840    // the same for the fp16 kernel (pss_xcorr_f16.hip)
#define LCS_MAX_WORK 1024    // most cells carried into the TFG/MIB stages per round (~6 MB each)
#define LCS_WORK_DEFAULT 512 // cells per round a context starts with (3 GB, allocated on first use); it doubles by itself when a batch carries more
// grid sizes of the work-list kernels (every one loops over its list, so these only trade latency for workgroups)
#define LCS_WIN_GRID 4096
#define LCS_ITEM_GRID 1024
#define LCS_TFG_GRID 4096
#define LCS_TFA_GRID 2048
#define LCS_TFG_ROWS 854
#define LCS_TFG_DESC_BYTES (856 * 32 + 128 * 16)   // tfg_mib.hip: TfgRow records + the frequency correction's position factors
#define LCS_CELL_SCRATCH 4608 // doubles of per-cell scratch (RS table, shifts, noise powers, PBCH candidates)

struct SlotParams {
  double fc_req, fc_prog, fs_prog;
};

struct XcGeom {
  uint32_t n_cap;
  int n_f;
  int n_tmpl;   // 3*n_f
  int cpg;      // template columns in use per 16-column group: 16 = dense packing, 15/12/9/6/3 = whole hypotheses per group
  int G;        // ceil(n_tmpl / cpg)
  int n_comb;   // n_comb_xc
  int ds;       // ds_comb_arm
  int foi0;     // hypothesis split over GPUs (lcs_foe_*): global index of this rank's first hypothesis; frq then holds GLOBAL indices
  int n_narrow; // windows 0 .. n_narrow - 1: the window starts of every template group (of every buffer of the call) lie within
                // LCS_NARROW_SPREAD samples of each other, so 137 taps + delay fit 144.  When that holds for every window of a call
                // (every grid the CLI builds) the fp16 kernel runs nine 16-tap blocks per window instead of ten (pss_xcorr_f16.hip);
                // the int8 kernel's half-depth last block was built and measured: no gain (profiles/r04/experiments)
};
#define LCS_NARROW_SPREAD 7

// int8 copies of a u8 capture buffer: slot stride in samples (a multiple of 8, so that every slot starts 16-byte
// aligned) with LCS_I8_PAD zero samples behind the data -- the correlation kernel's LDS-DMA reads run past n_cap
#define LCS_I8_PAD 1024
__host__ __device__ static inline size_t lcs_cap8_stride(uint32_t n_cap) { return (((size_t)n_cap + 7) & ~(size_t)7) + LCS_I8_PAD; }

// Template (foi * 3 + pss) held by column j of group g, or -1.  With the dense packing (cpg 16) consecutive templates fill
// the columns and a hypothesis may straddle two groups; a frequency grid too sparse for that (the window starts of the
// hypotheses in one group drift apart by more samples than the correlation kernels' tap blocks hold) is packed with
// fewer, whole hypotheses per group -- down to one (cpg 3), whose three templates share one window start.
__host__ __device__ static inline int lcs_col_tmpl(const XcGeom &geo, int g, int j) {
  const int c = g * geo.cpg + j;
  return (j < geo.cpg && c < geo.n_tmpl) ? c : -1;
}

// The capture buffers as the fp64 stages see them (exactly one pointer is set): the fp64 copy when a host entry
// point handed over complex<double>; the int8 pairs 127 - u8 of an RTL-SDR source ((u8-127)/128 = -a/128, exact);
// otherwise the fp32 copy widened on the fly (exact for float input).
struct CapSrc {
  const float2 *c32;
  const double2 *c64;
  const uint16_t *c8;
  uint32_t n_cap;
};
struct CapView {
  const float2 *c32;
  const double2 *c64;
  const uint16_t *c8;
};
#ifdef __HIPCC__
// Hand-over of LDS data between the lanes of ONE wave (the wave-local FFT stages): the wave barrier alone is IntrNoMem --
// no memory fence -- so the ordering of the LDS accesses around it is pinned by a release / acquire fence pair at
// wavefront scope (no instructions beyond the s_waitcnt the LDS reads need anyway).
__device__ __forceinline__ void lcs_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ CapView cap_view(const CapSrc &s, int slot) {
  CapView v;
  v.c32 = s.c32 ? s.c32 + (size_t)slot * s.n_cap : nullptr;
  v.c64 = s.c64 ? s.c64 + (size_t)slot * s.n_cap : nullptr;
  v.c8 = s.c8 ? s.c8 + (size_t)slot * lcs_cap8_stride(s.n_cap) : nullptr;
  return v;
}
__device__ __forceinline__ double2 cap_at(const CapView &v, size_t i) {
  if (v.c64) return v.c64[i];
  if (v.c8) {
    const uint32_t p = v.c8[i];
    return make_double2(-(double)(int)(int8_t)(p & 255u) / 128.0, -(double)(int)(int8_t)(p >> 8) / 128.0);
  }
  const float2 f = v.c32[i];
  return make_double2((double)f.x, (double)f.y);
}
#endif

// One "cell work item" for the per-cell stages (TFG / TFOEC / channel estimate / PBCH).
struct WorkItem {
  int slot;
  int peak;      // index into peaks[slot]
};

// Pinned block shared with the captured graph of the streaming mode.
// The first LCS_STREAM_IN_BYTES of it (parameters, tracked identities, the frequency hypothesis LAST) go to a device mirror of the same
// layout in ONE copy per push (rounds 2-5: four copy nodes per replay); the captured chain's kernels read their parameters and
// hypothesis from that mirror (stream_chain points the context's `params` / `fset` at it while the launches are recorded).
struct StreamHost {
  SlotParams p;
  int n_tracked;
  int pad_;
  int16_t tracked[504];
  double f;
  lcs_cell res[LCS_MAXP];
  int n_peaks;
  int n_work[4];
};

#define LCS_STREAM_IN_BYTES (offsetof(StreamHost, f) + sizeof(double))

struct lcs_ctx {
  int device = 0;
  hipStream_t stream = nullptr;      // everything except the PSS correlation (highest priority)
  hipStream_t stream_xc = nullptr;   // the PSS correlation kernel (lowest priority), see lcs_launch_xcorr
  hipEvent_t ev_pre = nullptr, ev_post = nullptr;
  std::string err;

  // capacity the workspace is currently sized for
  int cap_slots = 0;
  uint32_t cap_n_cap = 0;
  int cap_n_f = 0;
  int cap_G = 0;                     // template groups per buffer the tables are sized for
  bool cap_debug = false;

  // device buffers
  float2 *cap32 = nullptr;
  uint16_t *cap8 = nullptr;          // capture buffers as (re, im) int8 pairs 127 - u8, slot stride lcs_cap8_stride (pss_xcorr_i8.hip)
  uint16_t *cap8s = nullptr;         // the same shifted down by one sample: cap8s[i] = cap8[i + 1]
  uint32_t *brow8 = nullptr;         // int8 three-digit template operands: one image of resident rows per (slot, group)
  double *tq = nullptr;              // per template: integer scale q
  float *tsc = nullptr;              // per template: 1 / (128 q)
  bool i8_ready = false, use_i8 = false;
  // fp16 three-product path (pss_xcorr_f16.hip): complex<float> sources of the batched device entry points
  uint32_t *cap16h = nullptr, *cap16l = nullptr;   // (re, im) fp16 pairs, hi and lo parts, slot stride lcs_cap8_stride
  uint32_t *brow16 = nullptr;        // template operands, hi and lo terms: one image of resident rows per (slot, group)
  int *texp16 = nullptr;             // per template column: power-of-two scale exponent
  float *tsc16 = nullptr;            // per template column: 2^-(k_x + k_t)
  unsigned *xmax16 = nullptr;        // per slot: bits of the largest |component|
  unsigned *xpart16 = nullptr;       // per slot: 128 partial maxima (one per workgroup of the read-only maximum pass)
  bool f16_ready = false, use_f16 = false;
  bool src_u8 = false;               // the resident buffers came from a u8 source: the fp64 stages read cap8
  const float2 *src32 = nullptr;     // complex<float> batches read in place: the CALLER's buffers, which the fp64 stages read (no cap32 copy)
  double2 *cap64 = nullptr;          // slot 0 only: fp64 copy for the host (complex<double>) entry points
  bool cap64_valid = false;
  SlotParams *params = nullptr;
  double *fset = nullptr;
  float2 *tmpl = nullptr;
  int *start = nullptr, *smin = nullptr, *kp2 = nullptr;
  float *btab = nullptr;             // fp32 kernel only: allocated by its first launch for the workspace's slots and groups
  size_t btab_elems = 0;
  float *single = nullptr, *incoh = nullptr, *sref = nullptr;
  double *pow_ = nullptr, *work = nullptr, *spinc = nullptr, *zth = nullptr, *sp = nullptr;
  int *frq = nullptr;
  unsigned *fix_list = nullptr;      // [S][3][9600]: positions (slot * 3 + t) * 9600 + idx whose arg-max is a near-tie (capacity: every position)
  int *n_fix = nullptr;              // [4]: entries on the list (zeroed by k_prep_tables)
  float *second32 = nullptr;         // [S][3][9600]: the runner-up of the collapse's maximum (written for lcs_foe_partial only: lcs_foe_contend reads it)
  double *fset_g = nullptr;          // the whole grid, for lcs_foe_contend (fset holds the rank's share then)
  int fset_g_cap = 0;
  bool repair_peaks_only = false;    // lcs_search_capbuf / the streaming chain: list only the near-ties at or above their position's Z_th1 (pss_xcorr.hip: collapse_flag)
  bool skip_frq_repair = false;      // lcs_foe_partial: a rank sees only its share of the hypotheses (a near-tie may span two ranks)
  lcs_cell *peaks = nullptr;
  int *npeaks = nullptr;
  float2 *xc = nullptr;             // debug: raw correlations [3][n_cap-136][n_f]
  size_t xc_elems = 0;
  // SSS / FOE stage (sss_foe.hip): work list of (buffer, peak) pairs and per-(peak, occurrence) records
  WorkItem *pk_items = nullptr;
  int *n_pk = nullptr;
  double *sss_ws = nullptr;
  size_t sss_ws_items = 0;
  // per-cell stage buffers
  WorkItem *work_items = nullptr;
  int *n_work = nullptr;
  double2 *tfg = nullptr;           // [MAX_WORK][854][72]
  double2 *tfg_comp = nullptr;      // [MAX_WORK][854][72]
  double2 *ce = nullptr;            // [MAX_WORK][4][854][72]
  char *tfg_desc = nullptr;         // [MAX_WORK][LCS_TFG_DESC_BYTES]: per cell 856 window records in k_tfg's job order + 128 position factors (k_cell_prep)
  double *tfg_ts = nullptr;         // [MAX_WORK][854]
  double *tfg_ts_comp = nullptr;    // [MAX_WORK][854]
  double *cell_scratch = nullptr;   // [MAX_WORK][CELL_SCRATCH]
  lcs_cell *cells_out = nullptr;    // [MAX_WORK]
  // constant tables on the device
  double2 *d_pss_td = nullptr;      // [3][137]
  double2 *d_pss_fd = nullptr;      // [3][62]
  int8_t *d_sss_fd = nullptr;       // [168][3][2][62]
  uint8_t *d_pbch_scr = nullptr;    // [504][1920]
  uint32_t *d_pn_jump = nullptr;    // [32]: Gold-sequence jump-ahead by 1600 + 208 clocks (lcs_tables::pn_jump_table)
  int16_t *d_derm_inv = nullptr;    // [2][120][16]: for every coded bit (stream*40+col) the rate-matched PBCH bit positions carrying it (ascending, -1 padded)
  double *d_dbg = nullptr;          // debug outputs of the single-cell stage entry points
  int *d_flag = nullptr;            // exactness verdict of k_ingest_c128
  bool c64_probe = false;           // lcs_set_float_batch_probe: complex<float> batches are checked for dongle data (every component k/128) and then take the u8 route
  uint8_t *c64_u8 = nullptr;        // ... the bytes such a batch is turned into
  size_t c64_u8_bytes = 0;
  int c64_skip = 0;                 // batches left before the next probe (after a batch that was NOT dongle data)
  bool last_c64_routed = false;     // the last lcs_batch_enqueue of a complex<float> batch took the u8 route
  bool percell_ready = false;
  // streaming mode (lcs_stream_*): the one-buffer, n_f = 1 chain captured once as a hipGraph; every
  // per-push input reaches the device through fixed pinned buffers, so the graph never changes
  bool single_stream = false;        // launch everything on `stream`, no cross-stream events
  bool st_open = false;
  int st_head = 0, st_count = 0;                  // two slots: oldest buffer in flight, number in flight
  int st_fmt = 0;
  uint32_t st_n_cap = 0;
  void *st_hin[2] = {nullptr, nullptr};           // pinned copies of the pushed buffers
  void *st_din = nullptr;                         // device copy (shared: the graph launches serialise)
  size_t st_in_bytes = 0;
  struct StreamHost *st_host[2] = {nullptr, nullptr};   // pinned parameter + result blocks
  int16_t *st_dtracked = nullptr;     // (both point into st_dmirror)
  int *st_dntracked = nullptr;
  char *st_dmirror = nullptr;        // device mirror of StreamHost's input part
  hipGraph_t st_graph[2] = {nullptr, nullptr};
  hipGraphExec_t st_exec[2] = {nullptr, nullptr};
  hipEvent_t st_ev0[2] = {nullptr, nullptr}, st_ev1[2] = {nullptr, nullptr};
  // tracker block pipeline (tracker.hip): workspace laid out for one (n_cells, n_sym) block shape
  double2 *trk_td = nullptr, *trk_syms = nullptr, *trk_raw = nullptr, *trk_ce = nullptr;
  double *trk_meta = nullptr, *trk_rs = nullptr, *trk_fmeta = nullptr, *trk_pw = nullptr;
  int *trk_idx = nullptr, *trk_small = nullptr;
  lcs_track_cell *trk_cells = nullptr;
  int trk_cells_cap = 0, trk_sym_cap = 0;     // the workspace holds any block of up to this many cells x symbols
  int trk_last_cells = 0, trk_last_sym = 0;   // shape of the block the last lcs_track_block call processed (lcs_track_stats reads it)
  double2 *trk_acfd = nullptr, *trk_actd = nullptr, *trk_syncce = nullptr;   // lcs_track_stats outputs
  double *trk_sync = nullptr;
  int trk_stat_cells = 0, trk_stat_sym = 0;   // capacity of the statistics buffers
  void *trk_stream = nullptr;        // carried state of lcs_track_stream_block (tracker.hip)
  void *trk_hpin = nullptr;          // reusable host staging block of lcs_track_block (malloc): metadata up, measurement tables down
  size_t trk_hpin_bytes = 0;
  int *trk_cut_hit = nullptr;       // lcs_track_cut: first sample of every symbol [cells][symbols], per-cell flag / count behind it
  double *trk_cut_meta = nullptr;   // lcs_track_cut: late [cells][symbols], then frame_timing, freq_off [cells]
  size_t trk_cut_cap = 0;           // symbols x cells the two hold
  int trk_cut_cells_cap = 0;
  // results of a batch, compacted on the device (k_pack_results): [8 ints header][n_buf counts][records]; h_res = its page-locked mirror
  void *res_pack = nullptr, *h_res = nullptr;
  size_t res_pack_bytes = 0;
  double last_collect_host_us = 0;   // host time of the last lcs_batch_collect outside its wait for the GPU
  int collect_hint = 0;              // records the last collected batch returned: sizes the first copy of the next collect
  // host staging
  SlotParams h_params{};             // source of asynchronous parameter uploads of the single-buffer entry points
  void *h_pinned = nullptr;
  size_t h_pinned_bytes = 0;
  void *h2d = nullptr;               // device staging of lcs_batch_enqueue_host
  void *h_stage[2] = {nullptr, nullptr};        // pinned slots for host sources that are not page-locked
  hipEvent_t ev_stage[2] = {nullptr, nullptr};
  size_t h2d_bytes = 0;

  // last batch bookkeeping
  int last_n_buf = 0;
  int last_stage_mask = 0;
  int last_fmt = 0;
  bool needed_rows_only = false;     // fused chains: compute only the grid rows later stages read (tfg_mib.hip)
  int max_work = LCS_WORK_DEFAULT;   // cells per per-cell round (lcs_set_max_cells_in_flight; grows to LCS_MAX_WORK by itself unless the caller set it)
  bool max_work_pinned = false;      // the caller set the limit: it stays
  int percell_cap = 0;               // cells the per-cell buffers are allocated for
  int last_cell_rounds = 0;          // per-cell rounds launched for the last batch (round_cells cells each)
  int round_cells = LCS_WORK_DEFAULT; // max_work as it was when the last batch was enqueued
  int grid_items = 64;               // workgroups per work-list axis of the per-cell kernels (they loop over the list)
  int work_hint = 0;                 // cells the last collected batch carried into the per-cell stages: sizes the next batch's rounds and grids
  int hint_n_buf = 0, hint_fmt = -1, hint_stage = 0;   // the batch shape the hint was measured on
  XcGeom last_geo{};
  XcGeom foe_geo{};                  // lcs_foe_partial -> lcs_foe_finish: this rank's share of the hypotheses
  bool foe_ready = false;
  uint32_t foe_n_cap = 0;
  hipEvent_t ev_xc0 = nullptr, ev_xc1 = nullptr;
  int last_xc_launches = 0;
  double last_xc_ops = 0;            // matrix-core operations (2 x MACs) the correlation launches of the last batch executed
  const char *last_xc_kernel = "";
};

#ifndef LCS_EVENT_NOFENCE
#define LCS_EVENT_NOFENCE hipEventDisableSystemFence
#endif

#define HIPCHK(ctx, call)                                                        \
  do {                                                                           \
    hipError_t e_ = (call);                                                      \
    if (e_ != hipSuccess) {                                                      \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);            \
      return LCS_ERR_HIP;                                                        \
    }                                                                            \
  } while (0)

// ---- host tables (lte_tables.cpp) --------------------------------------------------
namespace lcs_tables {
void pss_fd(int n_id_2, double *re_im /*62*2*/);
void pss_td(int n_id_2, double *re_im /*137*2*/);
void sss_fd(int n_id_1, int n_id_2, int slot_num, int32_t *out /*62*/);
void lte_pn(uint32_t c_init, uint32_t len, uint8_t *out);
void pn_jump_table(uint32_t steps, uint32_t out[32]);
double chi2cdf_inv(double p, double k);
void pbch_deratematch_map(int n_e, uint8_t *out /*n_e*/);   // ref src/lte_lib.cpp:409-463 via :473-478
}  // namespace lcs_tables

void lcs_track_stream_free(lcs_ctx *c);   // tracker.hip

// ---- kernel launchers (one per .hip file) -------------------------------------------
// pss_xcorr.hip
CapSrc lcs_cap_src(const lcs_ctx *c, uint32_t n_cap);   // which copy of the capture buffers the fp64 stages read
int lcs_launch_ingest(lcs_ctx *c, const void *d_src, int fmt, int n_buf, uint32_t n_cap);
int lcs_launch_ingest_c128(lcs_ctx *c, uint32_t n_cap, bool *exact);   // cap64 -> cap32 + int8 copies; exact: every component is (u8 - 127) / 128
int lcs_launch_xcorr(lcs_ctx *c, int n_buf, const XcGeom &geo, bool want_incoh, bool time_it);
int lcs_ensure_btab(lcs_ctx *c);   // fp32 kernel's operand tables for the current workspace (allocated on first use)
int lcs_launch_single_layout(lcs_ctx *c, const XcGeom &geo, int slot, float *ref_layout, int to_ref);   // group-major <-> [t][idx][foi]
int lcs_launch_foe_contend(lcs_ctx *c, const XcGeom &geo, const double *fset_g, const long long *d_words, long long *d_words2);
int lcs_launch_foe_resolve(lcs_ctx *c, long long *d_words, const long long *d_words2);
// pss_xcorr_i8.hip
int lcs_launch_fill_brow_i8(lcs_ctx *c, int n_buf, const XcGeom &geo);
int lcs_launch_xcorr_i8(lcs_ctx *c, hipStream_t sxc, const XcGeom &geo, int slot0, int n_slots, int xcd_map);
// pss_xcorr_f16.hip
int lcs_launch_ingest_f16(lcs_ctx *c, const void *d_src, int n_buf, uint32_t n_cap);   // complex<float> -> cap32 + fp16 hi / lo pairs + per-buffer scale
int lcs_launch_fill_brow_f16(lcs_ctx *c, int n_buf, const XcGeom &geo);
int lcs_launch_xcorr_f16(lcs_ctx *c, hipStream_t sxc, const XcGeom &geo, int slot0, int n_slots, int xcd_map);

int lcs_launch_xc_debug(lcs_ctx *c, const XcGeom &geo);   // raw xc for slot 0 (debug output only)
// peak_search.hip
int lcs_launch_peak_search(lcs_ctx *c, int n_buf, const XcGeom &geo, double udb10_m12, bool fp32_exact);
int lcs_launch_foe_pack(lcs_ctx *c, const XcGeom &geo, long long *d_words, double *d_meta);      // collapsed (pow, frq) -> packed words
int lcs_launch_foe_unpack(lcs_ctx *c, const XcGeom &geo, const long long *d_words, const double *d_meta);
// sss_foe.hip
int lcs_launch_sss_foe(lcs_ctx *c, int n_buf, uint32_t n_cap, double thresh2_n_sigma, double *dbg /*device, nullable*/);
int lcs_launch_sss_only(lcs_ctx *c, uint32_t n_cap, double thresh2_n_sigma, double *dbg);
int lcs_launch_foe_only(lcs_ctx *c, uint32_t n_cap);
// tfg_mib.hip
int lcs_launch_gather_work(lcs_ctx *c, int n_buf, int skip /* cells already handled by earlier rounds */,
                           int limit = 0 /* cells of this round; 0: max_work */);
__host__ __device__ static inline size_t lcs_pack_rec_offset(int n_buf) { return ((size_t)(8 + n_buf) * sizeof(int) + 63) & ~(size_t)63; }
int lcs_launch_pack_results(lcs_ctx *c, int n_buf, bool full);   // peaks / npeaks (+ n_work) -> c->res_pack
int lcs_launch_rs_build(lcs_ctx *c);
int lcs_launch_tfg(lcs_ctx *c, uint32_t n_cap, bool with_rs /* also build RS_DL (the fused chain) */);
int lcs_launch_tfoec(lcs_ctx *c, bool apply_grid, int parts = 4 /* workgroups per cell for the timing estimate: 2 in batches */);
int lcs_launch_mib(lcs_ctx *c, bool fused);   // chan_est + PBCH candidates + selection (+ record back into the peak table)
int lcs_launch_chan_est(lcs_ctx *c);
void lcs_chan_est_np_layout(int *first, int *per_port, int *n_rs_first);   // where k_chan_est leaves its noise-power partial sums in cell_scratch
