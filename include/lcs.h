/*
 * lcs.h -- C ABI of the MI355X-native LTE cell-search hot path (liblcs_amd.so).
 *
 * Drop-in boundary for the searcher of Evrytania/LTE-Cell-Scanner: every entry point
 * below replaces one free function of the reference's include/searcher.h (cited per
 * function) with plain pointers and sizes -- no IT++ or torch types.  The C++ wrappers
 * in include/searcher_amd.h re-expose the reference's exact C++ signatures on top of
 * this ABI; INTEGRATION.md shows the binding a maintainer adds on the reference side.
 *
 * Conventions
 *  - complex arrays are interleaved (re, im); capture buffers are complex<double>
 *    exactly as itpp::cvec stores them (host entry points) or complex<float> / raw
 *    u8 I/Q already resident in HBM (device entry points).
 *  - 2-D outputs [3][9600] are row-major (PSS index major); the reference's itpp::mat is
 *    column-major, the C++ wrapper transposes.
 *  - 3-D outputs are [t][idx][foi] with foi fastest, identical to the reference's
 *    nested-vector vf3d / vcf3d (include/common.h.in:41-44).
 *  - every function returns LCS_OK (0) or a negative error; "not found" stays in-band in
 *    lcs_cell (n_id_1 == -1 / n_rb_dl == -1) exactly as in the reference
 *    (src/CellSearch.cpp:530, 554).
 *  - correlation kernel: the DATA decides.  Dongle samples are exactly (u8 - 127) / 128 (src/capbuf.cpp:172-181) and exact
 *    in int8: handed over as raw bytes (LCS_FMT_IQ_U8) or as complex<double> through the reference's call shape (the host
 *    entry points check every component on the device), they run on the int8 matrix cores (templates as 24-bit integers
 *    in three int8 digits, exact int32 accumulation); batches of complex<float> buffers (LCS_FMT_C64) run on the fp16
 *    matrix cores (fp16 hi + lo parts of samples and templates, three products, fp32 accumulation), any other single
 *    buffer on the fp32 MFMA kernel.  All agree with the reference to ~1e-6 relative or better.  Templates are processed 16 to a group; any f_search_set is accepted: a grid whose
 *    hypotheses' window starts drift apart by more samples than a group's operand rows hold (> 15 samples for int8 / fp16, > 111
 *    for fp32 -- far sparser than the 5 kHz grids of the CLI) is packed with fewer whole hypotheses per group, down to
 *    one, at proportionally more work.
 *  - a context owns one HIP device + stream + workspace; calls on one context are
 *    serialised by the caller, different contexts are independent (the reference's
 *    functions are re-entrant).
 *  - there is NO CPU fallback: if no gfx950 device is usable, lcs_create fails.
 */
#ifndef LCS_H
#define LCS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LCS_OK 0
#define LCS_ERR_NO_DEVICE (-1)
#define LCS_ERR_BAD_ARG (-2)
#define LCS_ERR_HIP (-3)
#define LCS_ERR_OVERFLOW (-4)   /* more results than the caller's array holds (outputs truncated) */
#define LCS_ERR_NOMEM (-5)

#define LCS_CP_UNKNOWN 0
#define LCS_CP_NORMAL 1
#define LCS_CP_EXTENDED 2

#define LCS_N_PSS 3
#define LCS_N_IDX 9600          /* 5 ms at 1.92 Msps */
#define LCS_TFG_NSC 72
#define LCS_TFG_MAX_OFDM 854    /* 6 frames + 2 slots, normal CP (src/searcher.cpp:895) */
/* The reference's peak_search appends to a std::list<Cell> without a bound (src/searcher.cpp:468-476).  The list is bounded
 * all the same: every peak zeroes 274 positions either side of itself in its PSS row (:480-485), so two peaks of one row lie
 * at least 275 positions apart on the circle of 9600 -- at most 34 per row, 102 per buffer -- and the loop ends at the
 * first maximum below Z_th1 (:449).  The library keeps LCS_MAX_PEAKS records per buffer: every list the reference can return
 * for a buffer whose thresholds are positive fits.  (A buffer of zeros has Z_th1 == 0 everywhere: the reference's loop then
 * never ends; here it stops at LCS_MAX_PEAKS peaks and the call reports LCS_ERR_OVERFLOW.) */
#define LCS_MAX_PEAKS 104
/* Footprint that follows from that bound: the SSS / FOE stage keeps 60 KB of window records per POSSIBLE peak of a batch
 * (n_buf x LCS_MAX_PEAKS x 8 half-frame occurrences x 3 KB -- 800 MB for a 128-buffer context, allocated by the first full-chain
 * batch), the peak table 104 x 104 B per buffer.  Real buffers carry a handful of peaks; the worst case is what is reserved. */

/* POD mirror of class Cell -- include/common.h.in:101-129, defaults src/common.cpp:36-56. */
typedef struct lcs_cell {
  double fc_requested;
  double fc_programmed;
  double pss_pow;
  double freq;
  double frame_start;
  double freq_fine;
  double freq_superfine;
  int32_t ind;
  int32_t n_id_2;
  int32_t n_id_1;
  int32_t cp_type;        /* LCS_CP_* */
  int32_t n_ports;
  int32_t n_rb_dl;
  int32_t phich_duration; /* 0 UNKNOWN, 1 NORMAL, 2 EXTENDED */
  int32_t phich_resource; /* 0 UNKNOWN, 1 oneSixth, 2 half, 3 one, 4 two */
  int32_t sfn;
  int32_t reserved;
} lcs_cell;

typedef struct lcs_ctx lcs_ctx;

/* ---- context ------------------------------------------------------------------- */
/* device < 0: use the current HIP device.  Fails (LCS_ERR_NO_DEVICE) without a GPU. */
int lcs_create(int device, lcs_ctx **out);
void lcs_destroy(lcs_ctx *ctx);
const char *lcs_last_error(const lcs_ctx *ctx);
const char *lcs_version(void);
void lcs_cell_init(lcs_cell *c);                      /* src/common.cpp:36-56 */
/* Memory-footprint limit: the per-cell stages (time-frequency grid, channel estimate, PBCH) hold at most n
 * detected cells at a time, ~6 MB each, allocated on first use FOR THAT MANY CELLS (n <= 1024).  Without this call a context
 * starts at 512 (3 GB) and doubles to 1024 by itself after a batch that carried more cells past SSS than that (a busy
 * band); after the call the limit stays where the caller put it.  A batch with more cells is processed in rounds:
 * lcs_batch_enqueue launches the rounds the PREVIOUS batch of the same shape needed (at least one detected cell per buffer),
 * lcs_batch_collect launches the rest if the device-side count says the batch had more.  Results do not depend on it and
 * no batch is truncated. */
int lcs_set_max_cells_in_flight(lcs_ctx *ctx, int n);
/* complex<float> BATCHES that are dongle data.  A float pipeline in front of the searcher often still carries RTL-SDR samples --
 * every component exactly (u8 - 127) / 128 (src/capbuf.cpp:172-181).  The host entry points recognise such data in complex<double> by
 * themselves and correlate the bytes (int8 kernel); for device-resident LCS_FMT_C64 batches the same check is opt-in, because it costs
 * lcs_batch_enqueue a host synchronisation: with on != 0 every such batch is first checked on the device (one pass that also writes the
 * bytes; the host waits for the verdict while the other contexts' kernels keep the GPU busy) and, if every component of every buffer
 * is on the 8-bit grid, takes the u8 route from there -- the same numbers through the int8 kernel (1.4 x the fp16 kernel's rate) and
 * the caller's buffer is not read again after the call returns; anything else takes the fp16 kernel as before (and the next fifteen
 * batches of the context are not checked).  Off by default.  Results do not depend on it. */
int lcs_set_float_batch_probe(lcs_ctx *ctx, int on);

/* ---- stage entry points (host buffers in / out) ------------------------------------ */

/* Replaces xcorr_pss -- include/searcher.h:22-41, src/searcher.cpp:389-419
 * (xc_correlate :113-174, xc_combine :263-308, xc_delay_spread :312-347, sp_est :185-221,
 * xc_peak_freq :353-383).  xc_re_im and sp are debug outputs of the reference and may be
 * NULL (raw xc is then never materialised); incoherent may be NULL. */
int lcs_xcorr_pss(lcs_ctx *ctx, const double *capbuf_re_im, uint32_t n_cap,
                  const double *f_search_set, uint16_t n_f, uint8_t ds_comb_arm,
                  double fc_requested, double fc_programmed, double fs_programmed,
                  double *xc_incoherent_collapsed_pow /*[3][9600]*/,
                  int32_t *xc_incoherent_collapsed_frq /*[3][9600]*/,
                  float *xc_incoherent_single /*[3][9600][n_f]*/,
                  float *xc_incoherent /*[3][9600][n_f] or NULL*/,
                  double *sp_incoherent /*[9600]*/,
                  float *xc_re_im /*[3][n_cap-136][n_f][2] or NULL*/,
                  double *sp /*[n_comb_sp*9600] or NULL*/,
                  uint16_t *n_comb_xc, uint16_t *n_comb_sp);

/* Replaces peak_search -- include/searcher.h:44-56, src/searcher.cpp:422-510.
 * Appends up to max_cells records; *n_cells receives the number found. */
int lcs_peak_search(lcs_ctx *ctx, const double *xc_incoherent_collapsed_pow,
                    const int32_t *xc_incoherent_collapsed_frq, const double *Z_th1 /*[9600]*/,
                    const double *f_search_set, uint16_t n_f, double fc_requested,
                    double fc_programmed, const float *xc_incoherent_single, uint8_t ds_comb_arm,
                    lcs_cell *cells, int max_cells, int *n_cells);

/* Replaces sss_detect -- include/searcher.h:59-76, src/searcher.cpp:696-761.  The eight
 * trailing outputs are the reference's "only used for testing" arrays; each may be NULL. */
int lcs_sss_detect(lcs_ctx *ctx, const lcs_cell *cell, const double *capbuf_re_im, uint32_t n_cap,
                   double thresh2_n_sigma, double fc_requested, double fc_programmed,
                   double fs_programmed, lcs_cell *cell_out,
                   double *sss_h1_np_est /*62*/, double *sss_h2_np_est /*62*/,
                   double *sss_h1_nrm_est /*62*2*/, double *sss_h2_nrm_est /*62*2*/,
                   double *sss_h1_ext_est /*62*2*/, double *sss_h2_ext_est /*62*2*/,
                   double *log_lik_nrm /*[168][2]*/, double *log_lik_ext /*[168][2]*/);

/* Replaces pss_sss_foe -- include/searcher.h:79-85, src/searcher.cpp:767-850. */
int lcs_pss_sss_foe(lcs_ctx *ctx, const lcs_cell *cell_in, const double *capbuf_re_im,
                    uint32_t n_cap, double fc_requested, double fc_programmed,
                    double fs_programmed, lcs_cell *cell_out);

/* Replaces extract_tfg -- include/searcher.h:88-98, src/searcher.cpp:857-935.
 * tfg is [n_ofdm][72] row-major; *n_ofdm = 854 (normal CP) or 732 (extended). */
int lcs_extract_tfg(lcs_ctx *ctx, const lcs_cell *cell, const double *capbuf_re_im, uint32_t n_cap,
                    double fc_requested, double fc_programmed, double fs_programmed,
                    double *tfg_re_im, double *tfg_timestamp, int *n_ofdm);

/* Replaces tfoec -- include/searcher.h:101-112, src/searcher.cpp:952-1069.  The RS_DL
 * argument of the reference is a pure function of (n_id_cell, cp_type) and is rebuilt
 * on the device (src/lte_lib.cpp:305-405). */
int lcs_tfoec(lcs_ctx *ctx, const lcs_cell *cell, const double *tfg_re_im,
              const double *tfg_timestamp, int n_ofdm, double fc_requested, double fc_programmed,
              double *tfg_comp_re_im, double *tfg_comp_timestamp, lcs_cell *cell_out);

/* Replaces decode_mib -- include/searcher.h:115-119, src/searcher.cpp:1526-1692
 * (chan_est :1369-1477, ce_interp_hex :1223-1362, pbch_extract :1482-1522). */
int lcs_decode_mib(lcs_ctx *ctx, const lcs_cell *cell, const double *tfg_re_im, int n_ofdm,
                   lcs_cell *cell_out);

/* chan_est (src/searcher.cpp:1369-1477 with ce_interp_hex :1223-1362), the first half of decode_mib, as a stage of its
 * own: channel estimate of antenna port `port` on the whole compensated grid and the port's noise power.  Internal to the
 * reference's searcher.cpp (not in searcher.h); exported so that it can be tested directly. */
int lcs_chan_est(lcs_ctx *ctx, const lcs_cell *cell, const double *tfg_re_im, int n_ofdm, int port,
                 double *ce_tfg_re_im /*[n_ofdm][72]*/, double *np);

/* ---- fused chain ------------------------------------------------------------------- */

/* One capture buffer through the whole chain of the reference's main loop
 * (src/CellSearch.cpp:484-558: xcorr_pss, Z_th1, peak_search, then per peak sss_detect,
 * pss_sss_foe, extract_tfg, tfoec, decode_mib, dropping peaks without SSS / MIB).
 * The buffer stays resident on the device across all stages.  peaks / n_peaks (nullable)
 * receive the raw peak_search list. */
int lcs_search_capbuf(lcs_ctx *ctx, const double *capbuf_re_im, uint32_t n_cap,
                      const double *f_search_set, uint16_t n_f, double fc_requested,
                      double fc_programmed, double fs_programmed,
                      lcs_cell *cells, int max_cells, int *n_cells,
                      lcs_cell *peaks, int max_peaks, int *n_peaks);

/* Batched, device-resident form used by sweeps and by bench.py: n_buf capture buffers of
 * n_cap samples each, ALREADY IN HBM, as complex<float> (fmt 0) or raw RTL-SDR u8 I/Q
 * bytes (fmt 1; (x-127)/128 is applied on the device, src/capbuf.cpp:172-181).
 * fc_requested / fc_programmed are per buffer (host arrays, n_buf).  cells is
 * [n_buf][max_cells_per_buf] on the host; n_cells [n_buf].  stage_mask selects how far the
 * chain runs: 1 = PSS correlation + peak_search only (BASELINE config 2), 3 = full chain. */
#define LCS_FMT_C64 0
#define LCS_FMT_IQ_U8 1
#define LCS_FMT_C128 2      /* complex<double>: lcs_track_cut only (the batch entry points take LCS_FMT_C64 / LCS_FMT_IQ_U8) */
#define LCS_STAGE_PSS 1
#define LCS_STAGE_FULL 3
int lcs_search_batch_dev(lcs_ctx *ctx, const void *d_capbufs, int fmt, int n_buf, uint32_t n_cap,
                         const double *f_search_set, uint16_t n_f, const double *fc_requested,
                         const double *fc_programmed, double fs_programmed, int stage_mask,
                         lcs_cell *cells, int max_cells_per_buf, int *n_cells);

/* The same for buffers in HOST memory (copied to the device by the call): what a caller holding recorded
 * capbuf_NNNN.it files or dongle bytes uses.  RTL-SDR captures are exactly (u8-127)/128 (src/capbuf.cpp:172-181):
 * handing them over as LCS_FMT_IQ_U8 moves 8x fewer bytes than complex<double> and takes the int8 correlation
 * kernel.  lcs_batch_enqueue_host returns as soon as the copy and the kernels are queued (results: lcs_batch_collect);
 * used round-robin over two or three contexts the PCIe transfer of batch i + 1 runs under the kernels of batch i.
 * The copy is a DMA straight from the caller's memory when that memory is page-locked (lcs_host_alloc); any other
 * pointer is staged through pinned slots inside the call (correct, but bounded by a CPU memcpy). */
int lcs_search_batch_host(lcs_ctx *ctx, const void *h_capbufs, int fmt, int n_buf, uint32_t n_cap,
                          const double *f_search_set, uint16_t n_f, const double *fc_requested,
                          const double *fc_programmed, double fs_programmed, int stage_mask,
                          lcs_cell *cells, int max_cells_per_buf, int *n_cells);
int lcs_batch_enqueue_host(lcs_ctx *ctx, const void *h_capbufs, int fmt, int n_buf, uint32_t n_cap,
                           const double *f_search_set, uint16_t n_f, const double *fc_requested,
                           const double *fc_programmed, double fs_programmed, int stage_mask);
/* Page-locked host memory for capture buffers (freed with lcs_host_free before the context is destroyed). */
int lcs_host_alloc(lcs_ctx *ctx, size_t bytes, void **out);
int lcs_host_free(lcs_ctx *ctx, void *p);
/* Device memory for callers that have no HIP toolchain of their own (the host tools are plain g++): buffers they hand to the
 * device-resident entry points (lcs_batch_enqueue, lcs_track_block with td_on_device, lcs_track_stream_block). */
int lcs_device_alloc(lcs_ctx *ctx, size_t bytes, void **out);
int lcs_device_free(lcs_ctx *ctx, void *p);
int lcs_device_upload(lcs_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);      /* synchronous host -> device copy */
/* Number of usable GPUs (0 without one): a sweep driver creates one context per device and shards the carriers
 * (host/CellSearch.cpp -g all; src/CellSearch.cpp:471-569 is the loop being sharded). */
int lcs_device_count(void);

/* Enqueue-only variant for timing: same work, results stay on the device until
 * lcs_batch_collect.  Between enqueue and collect nothing synchronises with the host.
 * The caller's device buffers must stay valid and unchanged until lcs_batch_collect has returned: u8 buffers are
 * converted by the first kernel of the chain, complex<float> buffers (even n_cap, 16-byte aligned) are read IN PLACE by
 * the SSS / FOE / grid stages as well -- the library keeps no float copy of them (lcs_batch_collect drops the reference
 * to them when it returns).  lcs_batch_collect copies only what the batch found: the records are compacted on the device,
 * and one small copy into page-locked memory of the context brings them over (no allocation inside the call). */
int lcs_batch_enqueue(lcs_ctx *ctx, const void *d_capbufs, int fmt, int n_buf, uint32_t n_cap,
                      const double *f_search_set, uint16_t n_f, const double *fc_requested,
                      const double *fc_programmed, double fs_programmed, int stage_mask);
int lcs_batch_collect(lcs_ctx *ctx, lcs_cell *cells, int max_cells_per_buf, int *n_cells);
/* Host time (microseconds) the last lcs_batch_collect of the context spent OUTSIDE its wait for the GPU: queueing the copy
 * and scattering the records into the caller's array (bench.py reports it as host_ms_per_batch.collect_excl_wait). */
int lcs_last_collect_host_us(lcs_ctx *ctx, double *us);
/* Counters of the last collected batch: stats[0] = records returned, [1] = 1 if a buffer overflowed LCS_MAX_PEAKS, [4] = cells the
 * last per-cell round took, [5] = peaks of the whole batch that passed sss_detect (the cells carried into extract_tfg .. decode_mib),
 * [6] = cells skipped as already tracked (streaming mode), [7] = PBCH candidates (frame timing x port count, 12 per cell:
 * src/searcher.cpp:1547, :1567) actually decoded -- the reference stops at the first candidate that passes, the batch kernel skips
 * a candidate whose cell already shows an earlier pass. */
int lcs_last_batch_stats(lcs_ctx *ctx, int stats[8]);
/* Debug readback of the last batch (after lcs_batch_collect / lcs_search_batch_*): the xcorr_pss outputs of
 * buffer `buf` in the layouts of lcs_xcorr_pss, plus the detection threshold Z_th1 (src/CellSearch.cpp:500-503).
 * Every pointer may be NULL.  This is how the tests pin the batched kernels to the oracle array by array. */
int lcs_batch_readback(lcs_ctx *ctx, int buf, float *xc_incoherent_single /*[3][9600][n_f]*/,
                       double *xc_incoherent_collapsed_pow /*[3][9600]*/, int32_t *xc_incoherent_collapsed_frq /*[3][9600]*/,
                       double *sp_incoherent /*[9600]*/, double *z_th1 /*[9600]*/);
/* xc_incoherent_collapsed_frq is an integer output of the reference (include/searcher.h:31): the first maximum over the
 * hypotheses of float values which the matrix-core kernels reproduce to ~1e-7, not to the bit.  Positions whose best two
 * hypotheses lie within 4e-6 (relative) of each other are therefore recomputed in the reference's own arithmetic (fp64
 * accumulation in tap order -> complex<float> -> float running sum over the windows -> float box filter, src/searcher.cpp:
 * 160-169, 299-305, 329-345) and decided with its strict comparison (:374): a few positions per buffer, rewritten in place
 * (index and power) before the peak search reads them.  *n_positions = how many the last correlation call of the context
 * repaired (all buffers of the batch).  lcs_search_capbuf and the streaming mode hand out no arrays, only peaks and cells: they
 * repair only the near-ties whose power reaches their position's threshold Z_th1 (a peak's power does, src/searcher.cpp:449) --
 * the peak list is exact all the same, and the latency of the repair stays off the single-buffer path.  lcs_foe_partial does not
 * repair (a near-tie may span two ranks' shares): lcs_foe_contend / lcs_foe_resolve settle those after the all-reduce. */
int lcs_last_frq_repairs(lcs_ctx *ctx, int *n_positions);
/* The repair's work is bounded: real captures list ~2 positions per buffer, but a degenerate input (duplicated entries of
 * f_search_set make every position an exact tie) lists all 3 x 9600.  Once a call lists more than 32 positions per repair
 * workgroup (256 for one buffer, 16384 for a 128-buffer batch) only positions whose power reaches their Z_th1 -- the ones a peak can
 * come from -- are recomputed, and each workgroup stops after 256 candidates; *n_unrepaired counts the listed positions left with
 * the correlation kernel's own arg-max (0 on any real data; for exact duplicates that arg-max is the reference's anyway:
 * identical templates give identical values and the first one wins). */
int lcs_last_frq_repair_stats(lcs_ctx *ctx, int *n_listed, int *n_unrepaired);
/* HIP-event time (ms) of the PSS correlation kernel launches of the last enqueue, and the
 * number of launches it covers; used by bench.py for the roofline figure. */
int lcs_last_xcorr_ms(lcs_ctx *ctx, float *ms, int *n_launches);
/* Which correlation kernel the last enqueue launched and the matrix-core operations (2 x multiply-accumulates,
 * padding included) those launches executed -- 0 when the kernel does not count them.  For bench.py's
 * roofline.frac_executed. */
int lcs_last_xcorr_info(lcs_ctx *ctx, double *executed_ops, const char **kernel);
/* ---- one capture buffer, frequency hypotheses split over GPUs ("latency mode") ----------------------------
 * a1 / a3 / a4 (src/searcher.cpp:113-347) are independent per hypothesis; the hypotheses meet only where xc_peak_freq takes
 * the maximum over the frequency axis (:369-382).  Every rank calls lcs_foe_partial with the whole f_search_set and its
 * contiguous share [f_first, f_first + f_count) (f_count may be 0); d_words (DEVICE memory of the caller, 3 x 9600 int64)
 * receives, per position, (bits(pow as float) << 32) | (0xFFFFFFFF - global hypothesis index): non-negative floats order
 * like their bit patterns and the complemented index makes the LOWEST hypothesis win a tie, as the reference's strict > does
 * (:374).  d_meta (DEVICE, 9601 doubles) receives sp_incoherent and n_comb_xc.  The caller all-reduces d_words with MAX
 * (RCCL: ncclMax on int64) and broadcasts rank 0's d_meta, both in place on the device, then every rank calls
 * lcs_foe_finish with the reduced buffers: peak_search runs identically everywhere; sss_detect .. decode_mib run for the
 * peaks whose winning hypothesis the rank owns (only it holds the xc_incoherent_single slice the refinement of `ind`
 * reads, :457-465).  cells / order: the rank's decoded cells and their positions in the peak list -- the caller gathers
 * them and sorts by `order` to get the reference's list.  peaks (nullable): the whole peak list; entries won by another
 * rank have ind = -1 and reserved = 1. */
int lcs_foe_partial(lcs_ctx *ctx, const double *capbuf_re_im, uint32_t n_cap, const double *f_search_set, uint16_t n_f,
                    int f_first, int f_count, double fc_requested, double fc_programmed, double fs_programmed,
                    void *d_words /*device int64[3][9600]*/, double *d_meta /*device double[9601]*/);
int lcs_foe_finish(lcs_ctx *ctx, const void *d_words, const double *d_meta, const double *f_search_set, uint16_t n_f,
                   lcs_cell *cells, int32_t *order, int max_cells, int *n_cells, lcs_cell *peaks, int max_peaks, int *n_peaks);
/* The exact index under the split (round 5), between the all-reduce of d_words and lcs_foe_finish:
 *   lcs_foe_contend(ctx, f_search_set, n_f, d_words [reduced], d_words2 [device int64[3][9600], out])
 *   MAX all-reduce of d_words2
 *   lcs_foe_resolve(ctx, d_words, d_words2)
 * A rank contends for a position when a hypothesis of its own other than the global winner lies within 4e-6 (relative) of the
 * global maximum; it recomputes its contenders and the winner in the reference's arithmetic (see lcs_last_frq_repairs) and packs
 * the exact first maximum into d_words2 (-1 where it does not contend).  After the second all-reduce and lcs_foe_resolve the words
 * -- index and power -- are the reference's at every near-tie, identically for every world size and every split of the grid.
 * Without these two calls the index is exact except at near-ties (the rounds before). */
int lcs_foe_contend(lcs_ctx *ctx, const double *f_search_set, uint16_t n_f, const void *d_words, void *d_words2);
int lcs_foe_resolve(lcs_ctx *ctx, void *d_words, const void *d_words2);

/* ---- streaming mode: LTE-Tracker's searcher thread (src/searcher_thread.cpp:83-246) ----------
 * One 80 ms capture buffer at a time, a single frequency hypothesis (the tracked frequency offset,
 * :97-98), cells whose identity is already tracked are reported as re-detected and not decoded again
 * (:157-177).  lcs_stream_open captures the whole chain as a hipGraph (twice: one graph per pinned input slot);
 * lcs_stream_push copies the HOST buffer (fmt LCS_FMT_C64: n_cap complex<float>, LCS_FMT_IQ_U8: 2*n_cap bytes) into
 * pinned memory and replays the graph asynchronously -- up to TWO buffers may be in flight, so the host fills and
 * launches buffer i + 1 while buffer i is on the GPU; lcs_stream_collect waits for the OLDEST buffer in flight and
 * returns its NEW cells (SSS and MIB decoded; one record per identity -- the first decoded peak in peak order, as the reference's
 * loop keeps it: it appends a decoded cell to the tracked list at once, :233-236, so later peaks of the same identity in the same
 * buffer count as seen again), the number of tracked cells seen again and the GPU time of the pass.
 * frame_start is in samples of the pushed buffer; the tracker's 1.92 MHz time base is
 * frame_start*(FS_LTE/16)/(fs_programmed*k_factor) + capture latency (:224).
 * While a stream is open the captured graph holds the context's workspace addresses: any other call on the
 * same context that would need a larger workspace (more buffers, a longer n_cap, more hypotheses) fails with
 * LCS_ERR_BAD_ARG until lcs_stream_close; use a second context for such work. */
int lcs_stream_open(lcs_ctx *ctx, int fmt, uint32_t n_cap, double fc_requested, double fc_programmed,
                    double fs_programmed);
int lcs_stream_push(lcs_ctx *ctx, const void *samples, double f_off, const int16_t *tracked_n_id_cell, int n_tracked);
int lcs_stream_collect(lcs_ctx *ctx, lcs_cell *cells, int max_cells, int *n_cells, int *n_redetected, float *gpu_ms);
int lcs_stream_close(lcs_ctx *ctx);

/* ---- LTE-Tracker's per-symbol pipeline on blocks of OFDM symbols (src/tracker_thread.cpp:823-1068) ------------
 * The reference tracks every detected cell with a thread that takes one OFDM symbol at a time: get_fd (:91-174), the
 * cell-specific reference symbols, filter_ce (:176-201) with its power measurements (:906-931), the frequency and
 * timing measurements of do_foe (:203-243) and do_toe_v2 (:245-288), interp2d (:383-477) and the MIB re-decode
 * (pbch_extract_rt :494-529, do_mib_decode :531-749).  lcs_track_block does that work for a BLOCK of n_sym
 * consecutive symbols (starting at slot 0 symbol 0 of a frame) of n_cells tracked cells at once.
 *   td            [n_cells][n_sym][128] complex<double>: the samples the producer thread queues per symbol
 *                 (src/producer_thread.cpp:196-246), in host memory or (td_on_device != 0) already in HBM
 *   freq_off, frame_timing, late [n_cells][n_sym]: the capture metadata queued with every symbol (host)
 *   cells         identity of each tracked cell; bulk_phase_offset is get_fd's running phase, in before / out after
 * Outputs (host; NULL = not wanted):
 *   syms          [n_cells][n_sym][72] complex: get_fd's output
 *   ce, ce_pw     [n_cells][4][n_sym][72] complex channel estimate and [n_cells][4][n_sym][4] (tp, sp, sp_raw, np) after
 *                 interp2d, valid for symbols < ce_upto[cell][port]
 *   meas          [n_cells][4][max_rs][LCS_TRK_MEAS]: per filtered reference symbol: symbol index, np, tp, sp_raw, sp,
 *                 frequency_offset + residual_f, residual_f_np (what do_foe feeds its recurrence, :235-242),
 *                 rs_curr.frame_timing + delay, delay_np (do_toe_v2, :283-287); n_meas [n_cells][4] rows filled
 *   mib_ok        [n_cells][max_off]: one decode attempt per frame offset o (frames o..o+3 of the block): bit 0 = CRC
 *                 matches, bit 1 = bandwidth / PHICH fields equal the tracked cell's (lock test of :689-694 = both), -1 = not
 *                 attempted (no channel estimate that far into the block); mib_bits: the 40 decoded bits, bit i = c_est(i)
 * The scalar recurrences those results feed (global frequency offset, frame timing, the mib_decode_failures counter)
 * stay with the caller (lte-cell-scanner_amd/tracker.py). */
typedef struct lcs_track_cell {
  int32_t n_id_1, n_id_2, cp_type, n_ports, n_rb_dl, phich_duration, phich_resource, reserved;
  double bulk_phase_offset;
} lcs_track_cell;
#define LCS_TRK_MEAS 9
int lcs_track_block(lcs_ctx *ctx, lcs_track_cell *cells, int n_cells, int n_sym, const void *td, int td_on_device,
                    const double *freq_off, const double *frame_timing, const double *late, double fc_requested,
                    double fc_programmed, double fs_programmed, double *syms, double *ce, double *ce_pw, int32_t *ce_upto,
                    double *meas, int max_rs, int32_t *n_meas, int32_t *mib_ok, uint64_t *mib_bits, int max_off, float *gpu_ms);

/* The producer thread's symbol extraction on the device (src/producer_thread.cpp:96-131, 196-246).  LTE-Tracker's producer stamps
 * every sample of the dongle's stream with a time on the cell-independent 1.92 MHz time base -- sample n of a buffer whose first
 * sample has timestamp ts_first: WRAP(ts_first + n step, 0, 19200), step = (FS_LTE/16) / (fs_programmed k_factor), k_factor =
 * (fc_requested - freq_off) / fc_programmed (:99, :127-131) -- and, per tracked cell, starts a 128-sample capture at the first sample
 * behind the previous capture whose tdiff = WRAP(timestamp - (frame_timing + target), -9600, 9600) satisfies |tdiff| < 0.5 or
 * 0 < tdiff < 3 (:203-213); target = 10 (normal CP) / 32 (extended) for slot 0 symbol 0, advancing by 137 / 138 / 160 per symbol
 * (:236-241); tdiff at that sample is the symbol's `late`.  lcs_track_cut does this for n_cells cells on ONE capture buffer that is
 * already in HBM -- 0.3 MB of dongle bytes per 80 ms instead of 2 KB per symbol and cell over PCIe -- and leaves the symbols where
 * lcs_track_block (td_on_device = 1) and lcs_track_stream_block read them:
 *   d_capbuf      DEVICE memory, n_cap samples: LCS_FMT_IQ_U8 (2 bytes per sample, (u8 - 127) / 128 as :121-124), LCS_FMT_C64,
 *                 LCS_FMT_C128
 *   ts_first      timestamp of the buffer's first sample (0 for a buffer cut from its start)
 *   cp_type, frame_timing, freq_off [n_cells] (host): the values in force while the buffer was recorded
 *   sym_first, pos_first [n_cells] (host; NULL = zeros): the first symbol to cut, counted from slot 0 symbol 0 of the stream's first
 *                 frame, and the sample of THIS buffer its search starts at
 *   d_td          DEVICE memory [n_cells][n_sym][128] complex<double>; rows from n_cut[cell] on are zero
 *   late          host [n_cells][n_sym] (may be NULL); n_cut [n_cells]: symbols found before the buffer ends (<= n_sym);
 *                 pos_next [n_cells] (may be NULL): the sample behind the last capture
 * A stream longer than one buffer: hand over the next buffer starting o samples into this one (o <= the smallest pos_next, so that a
 * capture the buffer's end cut off is whole in the next one) with ts_first' = WRAP(ts_first + o step, 0, 19200), sym_first' =
 * sym_first + n_cut, pos_first' = pos_next - o -- the producer's state between two blocks of samples, with the frame_timing and
 * frequency offset then in force.
 * Sample for sample what the host cutters produce (lte-cell-scanner_amd/tracker.py cut_symbols, host/TrackCells.cpp): the
 * window's position is known in closed form, the predicate itself is evaluated in the same double arithmetic on the candidates. */
int lcs_track_cut(lcs_ctx *ctx, const void *d_capbuf, int fmt, uint32_t n_cap, double ts_first, int n_cells, const int32_t *cp_type,
                  const double *frame_timing, const double *freq_off, const int64_t *sym_first, const int64_t *pos_first,
                  double fc_requested, double fc_programmed, double fs_programmed, int n_sym, void *d_td, double *late, int32_t *n_cut,
                  int64_t *pos_next);

/* Display statistics of the tracker thread for the block the LAST lcs_track_block call on this context processed (same
 * n_cells, n_sym; its symbols and raw reference-signal estimates are read from the workspace, nothing is recomputed):
 *   ac_fd   [n_cells][4][max_rs][12] complex: do_ac_fd (src/tracker_thread.cpp:318-341): autocorrelation over the 12 raw
 *           estimates of reference symbol row + 1 of the port (row = row of `meas`), divided by its signal power -- the
 *           value the reference then folds into tracked_cell.ac_fd with weight 1 / ac_fd_np (:335-338)
 *   ac_td   [n_cells][4][max_rs][72] complex: do_ac_td (:343-371): this_xc of the row against the 71 reference symbols
 *           before it, from row 71 on (NaN before: the reference's 72-deep history is not full)
 *   sync    [n_cells][max_hf][4] = (tp, sp, np, np_blank), sync_ce [n_cells][max_hf][72] complex: do_pss_sss_sigpower_ce
 *           (:754-820) of the k-th PSS/SSS pair of the block; n_hf [n_cells] pairs found
 * The running averages (ac_fd, ac_td, sync_*_av) are scalar recurrences and stay with the caller (tracker.py).
 * NULL = not wanted. */
int lcs_track_stats(lcs_ctx *ctx, int n_cells, int n_sym, double *ac_fd, double *ac_td, int max_rs, double *sync, double *sync_ce,
                    int max_hf, int32_t *n_hf);

/* Continuous form of lcs_track_block for callers that deliver a tracked cell's symbols block after block, as the reference's
 * producer thread does (src/producer_thread.cpp:196-246 -> the per-cell fifo read at src/tracker_thread.cpp:823-1068).  The
 * first call of a stream starts at slot 0 symbol 0 of a frame; every later call continues where the previous one ended
 * (any n_sym >= 1, the same cells in the same order).  Nothing is lost at the cut: the three-symbol window of filter_ce
 * (:176-201), the interpolation between filtered reference symbols (:383-477), the 72-deep history of do_ac_td (:343-371)
 * and the four-frame PBCH fifo (:552-745) all see the previous blocks' symbols, because the context carries the last 3-4
 * frames -- as frequency-domain rows (get_fd's output) on the device -- in front of the new symbols and runs the same
 * kernels over both (so every row is bit-identical to what ONE lcs_track_block call over the whole stream returns); the
 * carried frames are not transformed again, and frame offsets an earlier call attempted are not decoded again.  A call that
 * fails leaves the stream where it was.  Every output row is handed out exactly once, by the first call that can compute it,
 * under its index in the whole stream:
 *   syms      [n_cells][n_sym][72]: get_fd of the symbols of this call
 *   meas, ac_fd, ac_td [n_cells][4][max_rs][9 | 12 | 72 complex]: the filtered reference symbols that became available (a
 *             reference symbol's filter needs the NEXT one, so the last one of a block arrives with the next call); n_meas
 *             [n_cells][4] rows; meas[.][0] is the symbol index counted from the start of the stream
 *   ce, ce_pw [n_cells][4][ce_cap][72 complex | 4]: row r = symbol ce_from[cell][port] + r of the stream, ce_n rows
 *   mib_ok, mib_bits [n_cells][max_off]: entry k = frame offset mib_from[cell] + k of the stream (frames o..o+3), n_mib entries
 * cells[].bulk_phase_offset: in at the first call, out after every call.  td may be ordinary host memory, page-locked host
 * memory (lcs_host_alloc: DMA'd in place) or device memory (the kind is detected).  LIFETIME: page-locked and device memory
 * are copied asynchronously -- td must stay valid and unchanged until the call returns (the call synchronises with its stream
 * before it returns, so nothing of td is referenced afterwards); layout [n_cells][n_sym][128] complex<double>, rows exactly
 * n_sym * 128 elements apart.  lcs_track_stream_reset forgets the
 * stream (the next call starts a new one).  The PSS/SSS statistics (do_pss_sss_sigpower_ce) have no state across symbols
 * and stay with lcs_track_stats. */
int lcs_track_stream_block(lcs_ctx *ctx, lcs_track_cell *cells, int n_cells, int n_sym, const void *td, const double *freq_off,
                           const double *frame_timing, const double *late, double fc_requested, double fc_programmed,
                           double fs_programmed, double *syms, double *ce, double *ce_pw, int ce_cap, int64_t *ce_from, int32_t *ce_n,
                           double *meas, double *ac_fd, double *ac_td, int max_rs, int32_t *n_meas, int32_t *mib_ok,
                           uint64_t *mib_bits, int max_off, int64_t *mib_from, int32_t *n_mib);
int lcs_track_stream_reset(lcs_ctx *ctx);

/* Stream the context launches on (hipStream_t as void*), for external event timing. */
void *lcs_stream(lcs_ctx *ctx);
int lcs_sync(lcs_ctx *ctx);

/* ---- table accessors (tests compare them with the oracle) ---------------------------- */
int lcs_table_pss_td(int n_id_2, double *out_re_im /*137*2*/);    /* src/lte_lib.cpp:177-188 */
int lcs_table_pss_fd(int n_id_2, double *out_re_im /*62*2*/);     /* src/lte_lib.cpp:155-161 */
int lcs_table_sss_fd(int n_id_1, int n_id_2, int slot_num, int32_t *out /*62*/); /* :199-257 */
int lcs_table_lte_pn(uint32_t c_init, uint32_t len, uint8_t *out);               /* :41-147 */
double lcs_chi2cdf_inv(double p, double k);                       /* include/dsp.h:188-193 */

#ifdef __cplusplus
}
#endif
#endif /* LCS_H */
