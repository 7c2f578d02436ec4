// tfg_mib.hip -- per-cell stages: time/frequency grid, TFOEC, channel estimate, PBCH/MIB decode.
//
// Replaces extract_tfg (ref src/searcher.cpp:857-935), tfoec (:952-1069), chan_est (:1369-1477),
// ce_interp_hex (:1223-1362), pbch_extract (:1482-1522), decode_mib (:1526-1692) and the
// helpers they use from src/lte_lib.cpp (RS_DL :305-405, lte_demodulate :612-634,
// lte_conv_deratematch :469-518, lte_conv_decode :538-551, lte_calc_crc :637-663).
//
// All arithmetic is fp64 like the reference.  The cells that survived SSS detection in a whole
// batch of capture buffers are compacted into one work list; every kernel below is a fixed-size
// grid that strides over that device-side list, so the chain needs no host round trip.
#include "lcs_internal.h"

#define FS_LTE 30720000.0
#define N_RB_MAXDL 110
#define NSC 72
#define ROWS LCS_TFG_ROWS

// cell_scratch layout (doubles, per work item)
// Phase timestamps of workgroup (0,0,0) for tuning runs (-DLCS_PHASE_TS); compiled out otherwise.
#ifdef LCS_PHASE_TS
__device__ unsigned long long lcs_ph_ts[256];
#define PH(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) lcs_ph_ts[i] = wall_clock64(); } while (0)
extern "C" int lcs_debug_phase_ts(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lcs_ph_ts), sizeof(lcs_ph_ts)); }
#else
#define PH(i) do { } while (0)
#endif
#define CS_N_OFDM 0
#define CS_KFACTOR 1
#define CS_OOB 2
#define CS_NRS 8         // 4 values: RS rows per port
#define CS_NPP 640       // [4 ports][CE_NCHUNK <= 8] partial sums of |filtered - raw|^2
#define CS_TF_RES 680    // tfoec: residual_f, k_factor_residual, delay
#define CS_TF_KRES 681
#define CS_TF_TOE 684     // 4 complex partial sums of the timing estimate (k_tfoec_est parts), summed in part order
#define CS_CAND 16       // 12 candidates x 4: ok, bits lo (24 bits as double), unused
#define CS_SHIFT 64      // [140][4] (-1 = no RS)
#define CS_RS 1024       // [140][12] complex
#define CS_SIZE LCS_CELL_SCRATCH

#include "lte_device.h"

// block-wide sum of a complex value (any order; the reference sums sequentially, the
// difference is at the 1e-16 relative level)
__device__ cd2 block_sum(cd2 v, cd2 *red /*>= blockDim/64*/) {
  for (int off = 32; off > 0; off >>= 1) { v.re += __shfl_down(v.re, off); v.im += __shfl_down(v.im, off); }
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  cd2 s = mk(0, 0);
  for (int i = 0; i < nw; ++i) s = cadd(s, red[i]);
  __syncthreads();
  return s;
}

// ----------------------------------------------------------------- work-list compaction
// One wave: lane = capture buffer (64 at a time); cells that passed SSS are numbered in (buffer, peak)
// order by a prefix sum over the per-buffer counts; this launch takes numbers [skip, skip + limit).
// n_work[0] = cells taken, n_work[1] = cells that passed SSS in the whole batch (and are not already
// tracked), n_work[2] = cells skipped because their identity is in the tracked list (streaming mode,
// ref src/searcher_thread.cpp:157-177: a re-detected cell is not decoded again).
__device__ __forceinline__ bool cell_wanted(const lcs_cell &c, const int16_t *tracked, int n_tracked) {
  if (c.n_id_1 < 0) return false;
  const int id = c.n_id_2 + 3 * c.n_id_1;
  for (int i = 0; i < n_tracked; ++i) if (tracked[i] == id) return false;
  return true;
}
__global__ __launch_bounds__(64) void k_gather_work(const lcs_cell *__restrict__ peaks, const int *__restrict__ npeaks, int n_buf,
                                                    int skip, int limit, const int16_t *__restrict__ tracked,
                                                    const int *__restrict__ n_tracked_p, WorkItem *__restrict__ items,
                                                    int *__restrict__ n_work, lcs_cell *__restrict__ cells) {
  LCS_TAIL_PRIO();
  const int lane = threadIdx.x;
  const int n_tracked = tracked ? *n_tracked_p : 0;
  int base = 0, dup = 0;
  for (int s0 = 0; s0 < n_buf; s0 += 64) {
    const int s = s0 + lane;
    const int np = (s < n_buf) ? min(max(npeaks[s], 0), LCS_MAXP) : 0;
    int cnt = 0;
    for (int p = 0; p < np; ++p) {
      const lcs_cell c = peaks[(size_t)s * LCS_MAXP + p];
      const bool w = cell_wanted(c, tracked, n_tracked);
      cnt += w;
      dup += (c.n_id_1 >= 0 && !w);
    }
    int incl = cnt;
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if (lane >= off) incl += v; }
    int at = base + incl - cnt - skip;
    for (int p = 0; p < np; ++p) {
      const lcs_cell c = peaks[(size_t)s * LCS_MAXP + p];
      if (!cell_wanted(c, tracked, n_tracked)) continue;
      if (at >= 0 && at < limit) { items[at].slot = s; items[at].peak = p; cells[at] = c; }
      ++at;
    }
    base += __shfl(incl, 63);
  }
  for (int off = 32; off > 0; off >>= 1) dup += __shfl_down(dup, off);
  if (lane == 0) { n_work[0] = min(max(base - skip, 0), limit); n_work[1] = base; n_work[2] = dup; if (skip == 0) n_work[3] = 0; /* PBCH candidates decoded by the batch (k_pbch counts) */ }
}
// Results of a batch, compacted on the device for lcs_batch_collect: hdr[0] = records written, hdr[1] = 1 if a buffer
// found more peaks than LCS_MAXP holds (impossible for a buffer with positive thresholds, lcs.h), hdr[4..7] = n_work;
// cnt[b] = records of buffer b; rec = the records, buffer after buffer.  full: only peaks with SSS and MIB (the reference
// erases the others, src/CellSearch.cpp:530-534, :554-558); otherwise every peak.  The host then copies a few KB from
// ONE place instead of n_buf x LCS_MAXP records.
__global__ __launch_bounds__(64) void k_pack_results(const lcs_cell *__restrict__ peaks, const int *__restrict__ npeaks, int n_buf, int full,
                                                     const int *__restrict__ n_work, int *__restrict__ hdr, int *__restrict__ cnt,
                                                     lcs_cell *__restrict__ rec) {
  LCS_TAIL_PRIO();
  const int lane = threadIdx.x;
  int base = 0, ovf = 0;
  for (int s0 = 0; s0 < n_buf; s0 += 64) {
    const int s = s0 + lane;
    const int raw = (s < n_buf) ? max(npeaks[s], 0) : 0;
    const int np = min(raw, LCS_MAXP);
    ovf |= raw > LCS_MAXP;
    int c = 0;
    for (int p = 0; p < np; ++p) {
      const lcs_cell &pc = peaks[(size_t)s * LCS_MAXP + p];
      c += !(full && (pc.n_id_1 == -1 || pc.n_rb_dl == -1));
    }
    int incl = c;
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if (lane >= off) incl += v; }
    int at = base + incl - c;
    for (int p = 0; p < np; ++p) {
      const lcs_cell pc = peaks[(size_t)s * LCS_MAXP + p];
      if (full && (pc.n_id_1 == -1 || pc.n_rb_dl == -1)) continue;
      rec[at++] = pc;
    }
    if (s < n_buf) cnt[s] = c;
    base += __shfl(incl, 63);
  }
  ovf = __any(ovf);
  if (lane == 0) { hdr[0] = base; hdr[1] = ovf; hdr[2] = 0; hdr[3] = 0; }
  if (lane < 4) hdr[4 + lane] = n_work ? n_work[lane] : 0;
}
// ------------------------------------------------------------ extract_tfg: timestamps
// ref :875-889 and the running dft_location of :903-920 (kept sequential: each timestamp is a
// floating-point running sum).  One thread per cell.
__device__ void tfg_timestamps(const lcs_cell &c, const SlotParams &p, double *t_out, double *t_lds, double *sc) {
  const double k_factor = (p.fc_req - c.freq_fine) / p.fc_prog;
  const int n_symb = cell_n_symb(c);
  double loc;
  if (c.cp_type == LCS_CP_NORMAL) loc = c.frame_start + 10 * 16 / FS_LTE * p.fs_prog * k_factor;
  else loc = c.frame_start + 32 * 16 / FS_LTE * p.fs_prog * k_factor;
  if (loc - .01 * p.fs_prog * k_factor > -0.5) loc = loc - .01 * p.fs_prog * k_factor;
  const int n_ofdm = 6 * 10 * 2 * n_symb + 2 * n_symb;
  const double inc_ext = (128 + 32) * 16 / FS_LTE * p.fs_prog * k_factor;
  const double inc_10 = (128 + 10) * 16 / FS_LTE * p.fs_prog * k_factor;
  const double inc_9 = (128 + 9) * 16 / FS_LTE * p.fs_prog * k_factor;
  // one slot per iteration: the same running sum, without per-symbol bookkeeping in the dependent chain
  if (n_symb == 6) {
    for (int sl = 0; sl < n_ofdm / 6; ++sl) {
#pragma unroll
      for (int q = 0; q < 6; ++q) { t_out[sl * 6 + q] = loc; t_lds[sl * 6 + q] = loc; loc += inc_ext; }
    }
  } else {
    for (int sl = 0; sl < n_ofdm / 7; ++sl) {
#pragma unroll
      for (int q = 0; q < 7; ++q) { t_out[sl * 7 + q] = loc; t_lds[sl * 7 + q] = loc; loc += (q == 6) ? inc_10 : inc_9; }
    }
  }
  sc[CS_N_OFDM] = (double)n_ofdm;
  sc[CS_KFACTOR] = k_factor;
  sc[CS_OOB] = 0.0;
}

// One two-wave workgroup per cell prepares everything the grid kernels need: the first lane of
// wave 1 walks the OFDM symbol timestamps (mode bit 0) while lanes 0..59 of wave 0 build the
// cell's CRS table RS_DL (mode bit 1,
// ref src/lte_lib.cpp:305-383: values for the 6 centre RBs and the per-port frequency shifts of
// every (slot, symbol) that carries RS, one Gold sequence per lane).
// Round 5: wave 1 then also writes what a DFT window of k_tfg needs, one 32-byte record per window in the ORDER k_tfg takes
// them (TfgRow: the grid row, its ideal timestamp, and the frequency correction's factor for the window's first sample):
// extract_tfg rotates sample i of the buffer by cis(pi kk i), kk = -freq_fine / (fs / 2) (fshift, ref :892, dsp.h:40-53); for
// sample loc + n of a window that is cis(pi kk loc) cis(pi kk n) -- one factor per window and one per position (128 per
// cell, behind the records): 982 sincospi per cell here instead of one per sample in k_tfg (49 k per cell).
struct TfgRow { double2 rot; double ideal; int row; int pad; };
#define TFG_NDESC 856                                    // 107 jobs x 8 windows >= ROWS
#define TFG_DESC_BYTES LCS_TFG_DESC_BYTES
static_assert(sizeof(TfgRow) == 32 && TFG_DESC_BYTES == TFG_NDESC * 32 + 128 * 16, "descriptor block layout (lcs_internal.h)");
// The k-th row of the grid that the fused chain ever reads (tfg_row_needed below) -- slots carry 3 such rows, the PBCH slot of a
// frame 5 (normal CP) or 4 -- or -1 past the last one.
__device__ __forceinline__ int tfg_needed_row(int k, int n_symb, int n_ofdm) {
  const int extra = (n_symb == 7) ? 2 : 1, per_frame = 60 + extra;
  const int fr = k / per_frame, r = k - fr * per_frame;
  int slot, q;                                           // q-th needed row of the slot
  if (r < 3) { slot = 0; q = r; }
  else if (r < 6 + extra) { slot = 1; q = r - 3; }
  else { slot = 2 + (r - 6 - extra) / 3; q = (r - 6 - extra) % 3; }
  int sym;
  if (slot == 1) sym = q;                                // 0, 1, 2, 3 (, 4): n_symb - 3 is the last of them
  else sym = (q == 2) ? n_symb - 3 : q;
  const int row = (fr * 20 + slot) * n_symb + sym;
  return row < n_ofdm ? row : -1;
}
#define CP_THREADS 256
__global__ __launch_bounds__(CP_THREADS) void k_cell_prep(const lcs_cell *__restrict__ cells, const WorkItem *__restrict__ items,
                                                  const int *__restrict__ n_work, const SlotParams *__restrict__ params,
                                                  const uint32_t *__restrict__ pn_jump, double *__restrict__ ts,
                                                  double *__restrict__ scratch, char *__restrict__ desc, int mode, int needed_only) {
  LCS_TAIL_PRIO();
  __shared__ double s_ts[ROWS];
  const int tid = threadIdx.x;
  for (int it = blockIdx.x; it < *n_work; it += gridDim.x) {
    const lcs_cell c = cells[it];
    double *sc = scratch + (size_t)it * CS_SIZE;
    const SlotParams p = params[items[it].slot];
    const int n_symb = cell_n_symb(c), id = cell_id(c);
    if (tid >= 128 && tid < 140) sc[CS_CAND + (tid - 128) * 4] = 0.0;     // no PBCH candidate of this cell has passed yet (k_pbch's early exit reads these)
    // wave 1, lane 0: the timestamp walk; wave 0 meanwhile: the CRS table
    if ((mode & 1) && tid == 64) tfg_timestamps(c, p, ts + (size_t)it * ROWS, s_ts, sc);
    if ((mode & 2) && tid < 64 && n_symb >= 0 && id >= 0) {
      for (int e = tid; e < 140 * 4; e += 64) sc[CS_SHIFT + e] = -1.0;
      lcs_wave_sync();
      if (tid < 60) {      // one (slot, RS symbol) per lane
        const int slot = tid / 3, t = tid % 3;
        const int sym = (t == 2) ? (n_symb - 3) : t, row = slot * n_symb + sym;
        rs_dl_row(slot, t, id, c.cp_type, n_symb, pn_jump, &sc[CS_RS + row * 24], &sc[CS_SHIFT + row * 4]);
      }
    }
    __syncthreads();
    if ((mode & 1) && tid >= 64) {                       // waves 1-3: the window records and position factors
      const int n_ofdm = 6 * 10 * 2 * n_symb + 2 * n_symb;
      const double k_factor = (p.fc_req - c.freq_fine) / p.fc_prog;
      // fshift phase pi * (-f) / (fs/2) * n (ref dsp.h:40-53) as sincospi((-f)/(fs/2) * n): the absolute
      // sample index reaches 153600, far into the slow argument-reduction path of sincos
      const double kk = (-c.freq_fine) / ((p.fs_prog * k_factor) / 2);
      TfgRow *rd = reinterpret_cast<TfgRow *>(desc + (size_t)it * TFG_DESC_BYTES);
      double2 *pos = reinterpret_cast<double2 *>(rd + TFG_NDESC);
      for (int q = tid - 64; q < TFG_NDESC + 128; q += CP_THREADS - 64) {
        double sn, cs;
        if (q < TFG_NDESC) {
          int row = needed_only ? tfg_needed_row(q, n_symb, n_ofdm) : q;
          if (row >= n_ofdm) row = -1;
          const double ideal = row >= 0 ? s_ts[row] : 0.0;
          sincospi(kk * (double)d_round_i(ideal), &sn, &cs);
          TfgRow r;
          r.rot = make_double2(cs, sn); r.ideal = ideal; r.row = row; r.pad = 0;
          rd[q] = r;
        } else {
          sincospi(kk * (double)(q - TFG_NDESC), &sn, &cs);
          pos[q - TFG_NDESC] = make_double2(cs, sn);
        }
      }
    }
    __syncthreads();                                     // s_ts is rewritten by the next item
  }
}

// Rows of the time-frequency grid that the fused chain ever reads: the CRS symbols 0, 1, n_symb-3 of
// every slot (tfoec estimates, channel estimate) and PBCH symbols 0..3 of slot 1 of every frame
// (decode_mib) -- 45 % of the 854.  The stage entry points produce full grids (`needed_only` = 0).
__device__ __forceinline__ bool pbch_row(int t, int n_symb) {
  const int slot = t / n_symb, sym = t - slot * n_symb;
  return (slot % 20) == 1 && sym <= 3;
}
__device__ __forceinline__ bool tfg_row_needed(int t, int n_symb) {
  const int sym = t % n_symb;
  return sym == 0 || sym == 1 || sym == n_symb - 3 || pbch_row(t, n_symb);
}

// ------------------------------------------------------------------ extract_tfg: grid
// One wave per workgroup, 8 OFDM symbols (DFT windows) per job: lane (w, l) loads the 16 samples l + 8 j of window w straight
// from the capture buffer into registers, frequency-corrects them (the reference rotates all 153600 samples per cell, ref
// :892; only the windows that feed a DFT are touched here, with the same absolute-index phase, factored per window and per
// position by k_cell_prep), runs the register-resident 128-point transform (lte_device.h: fft128_x8), keeps the 72 occupied
// bins / sqrt(128) and applies the sub-sample timing phase ramp (ref :923-931).  Rounds 1-4: the windows lived in LDS through
// a fill pass and seven radix-2 stages (LDS-bound, and a sincospi per sample).  A job's dependent memory round trips are two:
// its window records (k_cell_prep wrote them in job order) and the work item, then the samples.
#define TFG_SYM 8
#define TFG_WAVES 4          // four independent waves per workgroup: in the pipelined chain a workgroup of this kernel starts where a
                             // correlation workgroup (4 waves, one per SIMD) retired -- as one-wave workgroups spread over the chip every
                             // one of them kept a whole correlation slot empty for the sake of one SIMD (measured: step - 9 %)
#define TFG_THREADS (64 * TFG_WAVES)
template <int KIND>      // which copy of the capture buffer: 0 = int8 pairs (dongle bytes), 1 = complex<float>, 2 = complex<double>
__device__ __forceinline__ void tfg_load16(const CapView &cap, long loc, int l, uint32_t n_cap, cd2 (&x)[16], bool &oob) {
  // sixteen loads in flight in the source's own width, no branch between them; converted afterwards
  uint16_t r8[16];
  float2 r32[16];
  unsigned in_mask = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const long sidx = loc + l + 8 * j;
    const bool in = sidx >= 0 && (uint64_t)sidx < n_cap;
    const size_t ci = in ? (size_t)sidx : 0;
    in_mask |= (in ? 1u : 0u) << j;
    if (KIND == 0) r8[j] = cap.c8[ci];
    else if (KIND == 1) r32[j] = cap.c32[ci];
    else { const double2 v = cap.c64[ci]; x[j] = mk(v.x, v.y); }
  }
  oob |= in_mask != 0xffffu;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (KIND == 0) { const uint32_t pr = r8[j]; x[j] = mk(-(double)(int)(int8_t)(pr & 255u) / 128.0, -(double)(int)(int8_t)(pr >> 8) / 128.0); }
    else if (KIND == 1) x[j] = mk((double)r32[j].x, (double)r32[j].y);
    if (!((in_mask >> j) & 1u)) x[j] = mk(0, 0);
  }
}
// (one instantiation per source format: with the three formats' load paths in one kernel the register allocation is the widest one's)
template <int KIND>
__global__ __launch_bounds__(TFG_THREADS) void k_tfg(const WorkItem *__restrict__ items, const int *__restrict__ n_work,
                                                     const CapSrc src, uint32_t n_cap, double *__restrict__ scratch,
                                                     const char *__restrict__ desc, double2 *__restrict__ tfg, int needed_only) {
  LCS_TAIL_PRIO();
  __shared__ cd2 tw[128];
  __shared__ cd2 tb_all[TFG_WAVES][TFG_SYM * FFT128_WSTRIDE];
  const int lane = threadIdx.x & 63, w = lane >> 3, l = lane & 7;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // wave-uniform, and the compiler is told so: the job's records load into SGPRs
  cd2 *tb = tb_all[wv];
  fft128_twiddle_table(tw, threadIdx.x, TFG_THREADS);
  __syncthreads();
  const int nw = *n_work;
  // a job = 8 consecutive rows (full grid), or 8 consecutive NEEDED rows (380 of the 854 with the normal CP: 48 jobs)
  const int jobs_per_item = needed_only ? 48 : TFG_NDESC / TFG_SYM;
  for (int job = blockIdx.x * TFG_WAVES + wv; job < nw * jobs_per_item; job += gridDim.x * TFG_WAVES) {
    const int it = job / jobs_per_item, jj = job % jobs_per_item;
    const TfgRow *rd = reinterpret_cast<const TfgRow *>(desc + (size_t)it * TFG_DESC_BYTES);
    const double2 *pos = reinterpret_cast<const double2 *>(rd + TFG_NDESC);
    const TfgRow d = rd[jj * TFG_SYM + w];
    const CapView cap = cap_view(src, items[it].slot);
    PH(30);
    const int row = d.row;
    cd2 x[16];
    bool oob = false;
    const long loc = (long)d_round_i(d.ideal);
    tfg_load16<KIND>(cap, loc, l, n_cap, x, oob);
    if (row >= 0) {
      // sample n = l + 8 j of the window is rotated by cis(pi kk (loc + n)): [window factor x position factor l] x position
      // factor 8 j, the latter as P(8 (j & 3)) P(32 (j >> 2)) -- six table values (uniform: scalar loads) instead of sixteen
      const cd2 c0 = cmul(mk(d.rot.x, d.rot.y), ld(&pos[l]));
      const cd2 a1 = ld(&pos[8]), a2 = ld(&pos[16]), a3 = ld(&pos[24]);
#pragma unroll
      for (int jh = 0; jh < 4; ++jh) {
        const cd2 ch = jh ? cmul(c0, ld(&pos[32 * jh])) : c0;
        x[4 * jh] = cmul(x[4 * jh], ch);
        x[4 * jh + 1] = cmul(x[4 * jh + 1], cmul(ch, a1));
        x[4 * jh + 2] = cmul(x[4 * jh + 2], cmul(ch, a2));
        x[4 * jh + 3] = cmul(x[4 * jh + 3], cmul(ch, a3));
      }
      if (oob) scratch[(size_t)it * CS_SIZE + CS_OOB] = 1.0;   // a sample of a DFT window outside the buffer: the reference would read out of bounds
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) x[j] = mk(0, 0);
    }
    PH(31);
    fft128_x8(x, tb, tw, lane);
    if (row >= 0) {
      const double late = (double)d_round_i(d.ideal) - d.ideal;      // |late| <= 1/2
      double k_im = -1.0;
      k_im = k_im * 2; k_im = k_im * M_PI; k_im = k_im * late; k_im = k_im / 128;
      // the timing phase ramp cis(k_im cn), cn = bin (1..36) or bin - 128 (-36..-1), bin = (l + 8 c) + 16 k1: cis(k_im (l + 8 c))
      // x cis(16 k_im m) with m = k1 or k1 - 8 in -3 .. 2; |k_im| <= pi / 128, so every argument below is < 0.4
      const cd2 e0 = cis_small_call(k_im * (double)l), g8 = cis_small_call(k_im * 8.0), e1 = cmul(e0, g8);     // cis(k_im (l + 8))
      const cd2 f1 = cmul(g8, g8), f2 = cmul(f1, f1), f3 = cmul(f2, f1);
      double2 *out = tfg + ((size_t)it * ROWS + row) * NSC;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int c = q >> 3, k1 = q & 7;
        if (k1 == 3 || k1 == 4) continue;                           // bins 48 .. 95: never kept
        const int bin = (l + 8 * c) + 16 * k1;                       // x[8 c + k1] = X[(l + 8 c) + 16 k1]
        const int i = (bin >= 92) ? bin - 92 : ((bin >= 1 && bin <= 36) ? bin + 35 : -1);
        if (i < 0) continue;
        const cd2 f = (k1 == 0) ? mk(1, 0) : (k1 == 1 ? f1 : (k1 == 2 ? f2 : (k1 == 5 ? cconj(f3) : (k1 == 6 ? cconj(f2) : cconj(f1)))));
        const cd2 a = cdivr(x[q], sqrt(128.0));
        st(&out[i], cmul(a, cmul(c ? e1 : e0, f)));
      }
    }
    PH(32);
  }
}

// ------------------------------------------------------------------------------ RS_DL accessors (table built by k_cell_prep)
__device__ __forceinline__ cd2 rs_val(const double *sc, int n_symb, int slot, int sym, int k) {
  const int row = slot * n_symb + sym;
  return mk(sc[CS_RS + (row * 12 + k) * 2], sc[CS_RS + (row * 12 + k) * 2 + 1]);
}
__device__ __forceinline__ int rs_shift(const double *sc, int n_symb, int slot, int sym, int port) {
  return (int)sc[CS_SHIFT + (slot * n_symb + sym) * 4 + port];
}

// ------------------------------------------------------------------------------ tfoec
// Two kernels: k_tfoec_est (one workgroup per cell) runs the two reductions -- the super-fine
// frequency estimate over the raw grid and the timing estimate over the frequency-corrected RS
// positions, correcting just those ~11k samples on the fly -- and k_tfoec_apply applies both
// corrections to the whole 854x72 grid with every element independent (grid = cells x row tiles) for the stage entry
// point lcs_tfoec; the fused chain does not materialise the corrected grid (k_chan_est applies the same expressions to
// the reference symbols and PBCH rows that are read afterwards).
// The value written for an element is (tfg * rot_f) * rot_late, then * rot_delay: the reference's
// order of the three complex products (ref :992-1005, :1061-1064).
// rot_f of one row: exp(j 2 pi (-residual_f) ts_comp / (FS_LTE/16)), the same for its 72 subcarriers
__device__ __forceinline__ cd2 foc_row_rot(double ts_t, double k_res, double residual_f) {
  const double tc = k_res * ts_t;
  double a_im = 1.0;
  a_im = a_im * 2; a_im = a_im * M_PI; a_im = a_im * (-residual_f); a_im = a_im * tc; a_im = a_im / (FS_LTE / 16);
  return cis_call(a_im);
}
__device__ __forceinline__ cd2 foc_value(const double2 *g, int t, int i, double ts_t, double k_res, cd2 rot_f) {
  const double tc = k_res * ts_t;
  const double late = ts_t - tc;
  double k_im = -1.0;
  k_im = k_im * 2; k_im = k_im * M_PI; k_im = k_im * late; k_im = k_im / 128;
  const double ph = k_im * (double)cn_of(i);
  const cd2 v = cmul(ld(&g[(size_t)t * NSC + i]), rot_f);
  return cmul(v, cis_auto_call(ph));
}

#define TF_THREADS 256
#define TF_PARTS 4           // at most: the timing-estimate sum of a cell is split over gridDim.y <= TF_PARTS workgroups (2 in batches -- every part repeats the
                             // frequency estimate, 4 cost a dense batch 1 % -- and 4 where one buffer's latency counts)
__global__ __launch_bounds__(TF_THREADS) void k_tfoec_est(lcs_cell *__restrict__ cells, const WorkItem *__restrict__ items,
                                                          const int *__restrict__ n_work,
                                                          const SlotParams *__restrict__ params,
                                                          const double2 *__restrict__ tfg, const double *__restrict__ ts,
                                                          double *__restrict__ scratch, double *__restrict__ ts_comp) {
  LCS_TAIL_PRIO();
  __shared__ cd2 red[TF_THREADS / 64];
  __shared__ cd2 rowrot[ROWS];
  const int tid = threadIdx.x, part_idx = blockIdx.y, n_parts = gridDim.y;     // the timing-estimate sum is split over n_parts workgroups
  for (int it = blockIdx.x; it < *n_work; it += gridDim.x) {
    const lcs_cell c = cells[it];
    const SlotParams p = params[items[it].slot];
    double *sc = scratch + (size_t)it * CS_SIZE;
    const int n_symb = cell_n_symb(c);
    const int n_ofdm = (int)sc[CS_N_OFDM];
    const int n_slot = n_ofdm / n_symb;
    const double2 *g = tfg + (size_t)it * ROWS * NSC;
    const double *tsi = ts + (size_t)it * ROWS;
    double *tsc = ts_comp + (size_t)it * ROWS;
    PH(10);
    // super-fine FOE (ref :970-989)
    cd2 part = mk(0, 0);
    for (int e = tid; e < 2 * 12 * (n_slot - 1); e += TF_THREADS) {
      const int r = e % (n_slot - 1), col = (e / (n_slot - 1)) % 12, st_ = e / ((n_slot - 1) * 12);
      const int sym = st_ ? (n_symb - 3) : 0;
      const int sh0 = rs_shift(sc, n_symb, d_imod(r, 20), sym, 0), sh1 = rs_shift(sc, n_symb, d_imod(r + 1, 20), sym, 0);
      const cd2 a = cmul(ld(&g[(size_t)(r * n_symb + sym) * NSC + sh0 + 6 * col]), cconj(rs_val(sc, n_symb, d_imod(r, 20), sym, col)));
      const cd2 b = cmul(ld(&g[(size_t)((r + 1) * n_symb + sym) * NSC + sh1 + 6 * col]), cconj(rs_val(sc, n_symb, d_imod(r + 1, 20), sym, col)));
      part = cadd(part, cmul(cconj(a), b));
    }
    const cd2 foe = block_sum(part, red);
    PH(11);
    const double residual_f = atan2(foe.im, foe.re) / (2 * M_PI) / 0.0005;
    const double k_res = (p.fc_req - residual_f) / p.fc_prog;
    for (int t = tid; t < n_ofdm; t += TF_THREADS) {
      if (part_idx == 0) tsc[t] = k_res * tsi[t];
      const int sym = t % n_symb;
      if (sym == 0 || sym == n_symb - 3) rowrot[t] = foc_row_rot(tsi[t], k_res, residual_f);   // the rows the TOE reads
    }
    __syncthreads();
    // TOE (ref :1012-1058) on the frequency-corrected RS samples
#define GC(row, col) foc_value(g, (row), (col), tsi[(row)], k_res, rowrot[(row)])
    part = mk(0, 0);
    const int n_toe = (2 * n_slot - 1) * 23, per_part = (n_toe + n_parts - 1) / n_parts;
    for (int e = part_idx * per_part + tid; e < min(n_toe, (part_idx + 1) * per_part); e += TF_THREADS) {
      const int t = e / 23, j = e % 23;
      const int cur_sym = (t & 1) ? (n_symb - 3) : 0, cur_slot = d_imod(t >> 1, 20), cur_off = (t >> 1) * n_symb + cur_sym;
      const int cur_sh = rs_shift(sc, n_symb, 0, cur_sym, 0);
      const int nxt_sym = ((t + 1) & 1) ? (n_symb - 3) : 0, nxt_slot = d_imod((t + 1) >> 1, 20), nxt_off = ((t + 1) >> 1) * n_symb + nxt_sym;
      const int nxt_sh = rs_shift(sc, n_symb, 0, nxt_sym, 0);
      int r1_off, r2_off, r1_sh, r2_sh, r1_sym, r2_sym, r1_slot, r2_slot;
      if (cur_sh < nxt_sh) { r1_off = cur_off; r1_sh = cur_sh; r1_sym = cur_sym; r1_slot = cur_slot; r2_off = nxt_off; r2_sh = nxt_sh; r2_sym = nxt_sym; r2_slot = nxt_slot; }
      else { r1_off = nxt_off; r1_sh = nxt_sh; r1_sym = nxt_sym; r1_slot = nxt_slot; r2_off = cur_off; r2_sh = cur_sh; r2_sym = cur_sym; r2_slot = cur_slot; }
      if (j < 12) {   // toe1: conj(r1v[j]) * r2v[j]
        const cd2 r1 = cmul(GC(r1_off, r1_sh + 6 * j), cconj(rs_val(sc, n_symb, r1_slot, r1_sym, j)));
        const cd2 r2 = cmul(GC(r2_off, r2_sh + 6 * j), cconj(rs_val(sc, n_symb, r2_slot, r2_sym, j)));
        part = cadd(part, cmul(cconj(r1), r2));
      } else {        // toe2: conj(r2v[i]) * r1v[i+1], i = 0..10
        const int i = j - 12;
        const cd2 r2 = cmul(GC(r2_off, r2_sh + 6 * i), cconj(rs_val(sc, n_symb, r2_slot, r2_sym, i)));
        const cd2 r1 = cmul(GC(r1_off, r1_sh + 6 * (i + 1)), cconj(rs_val(sc, n_symb, r1_slot, r1_sym, i + 1)));
        part = cadd(part, cmul(cconj(r2), r1));
      }
    }
#undef GC
    const cd2 toe = block_sum(part, red);
    PH(13);
    if (tid == 0) {
      sc[CS_TF_TOE + 2 * part_idx] = toe.re;
      sc[CS_TF_TOE + 2 * part_idx + 1] = toe.im;
      if (part_idx == 0) {
        for (int q = n_parts; q < TF_PARTS; ++q) { sc[CS_TF_TOE + 2 * q] = 0.0; sc[CS_TF_TOE + 2 * q + 1] = 0.0; }      // the consumers add TF_PARTS partial sums
        sc[CS_TF_RES] = residual_f;
        sc[CS_TF_KRES] = k_res;
        cells[it].freq_superfine = c.freq_fine + residual_f;
      }
    }
    __syncthreads();
    PH(14);
  }
}

// the pieces of the correction that k_tfoec_est leaves in the cell's scratch, as every consumer forms them
struct TfoecCorr { double residual_f, k_res, delay; };
__device__ __forceinline__ TfoecCorr tfoec_corr(const double *sc) {
  TfoecCorr r;
  r.residual_f = sc[CS_TF_RES];
  r.k_res = sc[CS_TF_KRES];
  cd2 toe = mk(0, 0);
  for (int q = 0; q < TF_PARTS; ++q) toe = cadd(toe, mk(sc[CS_TF_TOE + 2 * q], sc[CS_TF_TOE + 2 * q + 1]));
  r.delay = -atan2(toe.im, toe.re) / 3 / (2 * M_PI / 128);          // ref :1058
  return r;
}
// per-subcarrier rotation of the timing correction (ref :1061-1064)
__device__ __forceinline__ cd2 toc_subcarrier_rot(double delay, int i) {
  double k_im = 1.0;
  k_im = k_im * 2; k_im = k_im * M_PI; k_im = k_im / 128; k_im = k_im * delay;
  const double ph = k_im * (double)cn_of(i);
  return cis_auto_call(ph);
}

#define TFA_ROWS 8
#define TFA_THREADS 192
__global__ __launch_bounds__(TFA_THREADS) void k_tfoec_apply(const int *__restrict__ n_work, const double2 *__restrict__ tfg,
                                                             const double *__restrict__ ts, const double *__restrict__ scratch,
                                                             const lcs_cell *__restrict__ cells, double2 *__restrict__ tfg_comp,
                                                             int needed_only) {
  LCS_TAIL_PRIO();
  __shared__ cd2 comp[NSC];
  __shared__ cd2 rowrot[TFA_ROWS];
  const int tid = threadIdx.x;
  const int tiles = (ROWS + TFA_ROWS - 1) / TFA_ROWS;
  int comp_it = -1;
  for (int job = blockIdx.x; job < *n_work * tiles; job += gridDim.x) {
    const int it = job / tiles, t0 = (job % tiles) * TFA_ROWS;
    const double *sc = scratch + (size_t)it * CS_SIZE;
    const int n_ofdm = (int)sc[CS_N_OFDM];
    const TfoecCorr corr = tfoec_corr(sc);
    const double residual_f = corr.residual_f, k_res = corr.k_res, delay = corr.delay;
    const double2 *g = tfg + (size_t)it * ROWS * NSC;
    double2 *gc = tfg_comp + (size_t)it * ROWS * NSC;
    const double *tsi = ts + (size_t)it * ROWS;
    const int n_symb = cell_n_symb(cells[it]);
    if (it != comp_it) {     // per-subcarrier rotation of the timing correction (ref :1061-1064)
      __syncthreads();
      if (tid < NSC) comp[tid] = toc_subcarrier_rot(delay, tid);
      __syncthreads();
      comp_it = it;
    }
    __syncthreads();
    if (tid < TFA_ROWS && t0 + tid < n_ofdm) rowrot[tid] = foc_row_rot(tsi[t0 + tid], k_res, residual_f);
    __syncthreads();
    for (int e = tid; e < TFA_ROWS * NSC; e += TFA_THREADS) {
      const int t = t0 + e / NSC, i = e % NSC;
      if (t < n_ofdm && (!needed_only || tfg_row_needed(t, n_symb)))
        st(&gc[(size_t)t * NSC + i], cmul(foc_value(g, t, i, tsi[t], k_res, rowrot[e / NSC]), comp[i]));
    }
  }
}

// --------------------------------------------------------------------------- chan_est
// 3x3 complex solve by LU with partial pivoting (IT++ inv -> LAPACK zgetrf/zgetri in the reference)
__device__ void solve3(cd2 M[3][3], cd2 V[3], cd2 out[3]) {
  cd2 A[3][4];
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) A[i][j] = M[i][j]; A[i][3] = V[i]; }
  for (int col = 0; col < 3; ++col) {
    int piv = col;
    double best = fabs(A[col][col].re) + fabs(A[col][col].im);
    for (int r = col + 1; r < 3; ++r) { const double v = fabs(A[r][col].re) + fabs(A[r][col].im); if (v > best) { best = v; piv = r; } }
    if (piv != col) for (int j = 0; j < 4; ++j) { const cd2 t = A[col][j]; A[col][j] = A[piv][j]; A[piv][j] = t; }
    for (int r = col + 1; r < 3; ++r) {
      const cd2 f = cdiv(A[r][col], A[col][col]);
      for (int j = col; j < 4; ++j) A[r][j] = csub(A[r][j], cmul(f, A[col][j]));
    }
  }
  for (int i = 2; i >= 0; --i) {
    cd2 s = A[i][3];
    for (int j = i + 1; j < 3; ++j) s = csub(s, cmul(A[i][j], out[j]));
    out[i] = cdiv(s, A[i][i]);
  }
}
// ref include/dsp.h:151-185 interp1 (bisection with round_i midpoint, linear, extrapolating)
__device__ cd2 interp1_c(const double *X, const cd2 *Y, int n, double x) {
  if (n == 1) return Y[0];
  unsigned l = 0, r = (unsigned)n - 1;
  while (r - l > 1) {
    const unsigned mid = (unsigned)d_round_i((r + l) / 2.0);
    if (x >= X[mid]) l = mid; else r = mid;
  }
  const cd2 d = csub(Y[r], Y[l]);
  return cadd(Y[l], cdivr(cscale(d, (x - X[l])), (X[r] - X[l])));
}
// ref :1200-1213
__device__ int hex_extend(double *row_x, cd2 *row_val, int len) {
  if (row_x[0] != 0) {
    const cd2 d = csub(row_val[1], row_val[0]);
    const cd2 v = csub(row_val[0], cdivr(cscale(d, row_x[0]), (row_x[1] - row_x[0])));
    for (int i = len; i > 0; --i) { row_val[i] = row_val[i - 1]; row_x[i] = row_x[i - 1]; }
    row_val[0] = v; row_x[0] = 0; ++len;
  }
  if (row_x[len - 1] != 71) {
    const cd2 d = csub(row_val[len - 1], row_val[len - 2]);
    const cd2 v = cadd(row_val[len - 1], cdivr(cscale(d, (71 - row_x[len - 1])), (row_x[len - 1] - row_x[len - 2])));
    row_val[len] = v; row_x[len] = 71; ++len;
  }
  return len;
}

// Vertex i of the (edge-extended) RS row with first RS subcarrier `s` (ref :1200-1213): the 12
// filtered estimates at s, s+6, ..., plus a linearly extrapolated vertex at subcarrier 0 when
// s != 0 and at 71 when the row does not end there.
__device__ __forceinline__ int ext_len(int s) { return 12 + (s != 0) + (s != 5); }
__device__ __forceinline__ void ext_vertex(const cd2 *row, int s, int i, int &x, cd2 &v) {
  const int lead = (s != 0);
  const int j = i - lead;
  if (j < 0) {                       // left edge: row_val(0)-row_x(0)*(row_val(1)-row_val(0))/(row_x(1)-row_x(0))
    const cd2 d = csub(row[1], row[0]);
    v = csub(row[0], cdivr(cscale(d, (double)s), 6.0));
    x = 0;
  } else if (j > 11) {               // right edge
    const cd2 d = csub(row[11], row[10]);
    v = cadd(row[11], cdivr(cscale(d, (double)(71 - (s + 66))), 6.0));
    x = 71;
  } else { v = row[j]; x = s + 6 * j; }
}

// four waves per workgroup: the filter phases use all of them, the interpolation gives every output row a wave (one
// lane per triangle of its strip)
#define CE_THREADS 256
// RS row list of one port (ref :1383-1392) in closed form: ports 0/1 carry RS in symbols 0 and
// n_symb-3 of every slot (the sorted union alternates between the two), ports 2/3 in symbol 1.
__device__ __forceinline__ int ce_rs_row(int port, int n_symb, int t) {
  return (port <= 1) ? (t >> 1) * n_symb + ((t & 1) ? n_symb - 3 : 0) : 1 + t * n_symb;
}
__device__ __forceinline__ int ce_rs_count(int port, int n_symb, int n_ofdm) {
  if (port <= 1) return (n_ofdm - 1) / n_symb + 1 + ((n_ofdm - 1 >= n_symb - 3) ? (n_ofdm - 1 - (n_symb - 3)) / n_symb + 1 : 0);
  return (n_ofdm - 1 >= 1) ? (n_ofdm - 2) / n_symb + 1 : 0;
}
#define CE_MAX_RS 256
#define CE_CH 48                                      // RS rows per workgroup: 19 KB of LDS
#define CE_NCHUNK ((CE_MAX_RS + CE_CH - 1) / CE_CH)   // 6 (scratch holds 8 partials per port)
// In the fused chain (tfg_raw != nullptr) the grid has NOT been through k_tfoec_apply: the frequency and timing corrections
// (ref :992-1005, :1061-1064) are applied here, with k_tfoec_apply's expressions, to the values this kernel reads -- the
// reference symbols of the port, 12 of a row's 72 subcarriers -- and the port-0 workgroups write the corrected PBCH rows
// (4 per frame: all that k_pbch reads of the grid) into tfg_comp.  Round 2 corrected 390 rows x 72 subcarriers per cell in
// a kernel of its own, three sincos pairs per element, to have 11 k of the 28 k values read back here.
__global__ __launch_bounds__(CE_THREADS) void k_chan_est(const lcs_cell *__restrict__ cells, const int *__restrict__ n_work,
                                                          double2 *__restrict__ tfg_comp, const double2 *__restrict__ tfg_raw,
                                                          const double *__restrict__ ts,
                                                          double *__restrict__ scratch, double2 *__restrict__ ce, int pbch_only) {
  LCS_TAIL_PRIO();
  __shared__ cd2 ce_raw[(CE_CH + 3) * 12];     // RS rows r0 .. r1 of this chunk (one halo row each side)
  __shared__ cd2 ce_filt[(CE_CH + 1) * 12];    // RS rows c0 .. f1
  __shared__ cd2 red[CE_THREADS / 64];
  __shared__ cd2 toc_rot[NSC];                 // fused chain: per-subcarrier rotation of the timing correction
  __shared__ cd2 foc_rot[CE_CH + 3];           // fused chain: per-row rotation of the frequency correction, rows r0 .. r1
  const int tid = threadIdx.x, port = blockIdx.y, chunk = blockIdx.z;
  for (int it = blockIdx.x; it < *n_work; it += gridDim.x) {
    const lcs_cell c = cells[it];
    double *sc = scratch + (size_t)it * CS_SIZE;
    const int n_symb = cell_n_symb(c);
    const int n_ofdm = (int)sc[CS_N_OFDM];
    const double2 *g = tfg_comp + (size_t)it * ROWS * NSC;
    const double2 *graw = tfg_raw ? tfg_raw + (size_t)it * ROWS * NSC : nullptr;
    const double *tsi = tfg_raw ? ts + (size_t)it * ROWS : nullptr;
    TfoecCorr corr = {0.0, 0.0, 0.0};
    double2 *out = ce + (((size_t)it * 4 + port) * ROWS) * NSC;
    __syncthreads();
    PH(19);
    if (graw) {
      corr = tfoec_corr(sc);
      if (tid < NSC) toc_rot[tid] = toc_subcarrier_rot(corr.delay, tid);
      __syncthreads();
      if (port == 0) {          // the PBCH rows of frame f (slot 20 f + 1, symbols 0..3), frames dealt round-robin to the chunks
        double2 *gw = tfg_comp + (size_t)it * ROWS * NSC;
        for (int f = chunk; (20 * f + 1) * n_symb < n_ofdm; f += CE_NCHUNK) {
          __syncthreads();
          if (tid < 4 && (20 * f + 1) * n_symb + tid < n_ofdm)                      // the row's frequency rotation once, not per element
            foc_rot[tid] = foc_row_rot(tsi[(20 * f + 1) * n_symb + tid], corr.k_res, corr.residual_f);
          __syncthreads();
          for (int e = tid; e < 4 * NSC; e += CE_THREADS) {
            const int t = (20 * f + 1) * n_symb + e / NSC, i = e % NSC;
            if (t < n_ofdm)
              st(&gw[(size_t)t * NSC + i], cmul(foc_value(graw, t, i, tsi[t], corr.k_res, foc_rot[e / NSC]), toc_rot[i]));
          }
        }
        __syncthreads();
      }
    }
#define rs_set(t) ce_rs_row(port, n_symb, (t))
    const int n_rs = min(ce_rs_count(port, n_symb, n_ofdm), CE_MAX_RS);
    // slot_num advances every 2nd RS row for ports 0/1, every row for ports 2/3 (quirk Q9)
    const int sh0 = rs_shift(sc, n_symb, 0, d_imod(rs_set(0), n_symb), port);
    const int sh1 = rs_shift(sc, n_symb, d_imod((port >= 2) ? 1 : 0, 20), d_imod(rs_set(1), n_symb), port);
    PH(20);
    // this workgroup: filtered rows and row pairs [c0, c1) of the RS row list
    const int c0 = chunk * CE_CH, c1 = min(n_rs, c0 + CE_CH);
    if (c0 >= n_rs) {
      if (tid == 0) sc[CS_NPP + port * 8 + chunk] = 0.0;
      continue;
    }
    const int f1 = min(c1, n_rs - 1);                          // last filtered row needed (inclusive)
    const int r0 = max(c0 - 1, 0), r1 = min(f1 + 1, n_rs - 1);  // raw rows needed (inclusive)
    if (graw) {
      for (int t = r0 + tid; t <= r1; t += CE_THREADS) foc_rot[t - r0] = foc_row_rot(tsi[rs_set(t)], corr.k_res, corr.residual_f);
      __syncthreads();
    }
    for (int e = tid; e < (r1 - r0 + 1) * 12; e += CE_THREADS) {
      const int t = r0 + e / 12, i = e % 12;
      const int slot = (port >= 2) ? (t % 20) : ((t >> 1) % 20);
      const int sym = d_imod(rs_set(t), n_symb);
      const int sh = rs_shift(sc, n_symb, slot, sym, port);
      const int row = rs_set(t), col = sh + 6 * i;
      const cd2 v = graw ? cmul(foc_value(graw, row, col, tsi[row], corr.k_res, foc_rot[t - r0]), toc_rot[col])
                         : ld(&g[(size_t)row * NSC + col]);
      ce_raw[e] = cmul(v, cconj(rs_val(sc, n_symb, slot, sym, i)));
    }
    __syncthreads();
    PH(21);
    // 7-point hexagonal mean (ref :1431-1467)
    for (int e = tid; e < (f1 - c0 + 1) * 12; e += CE_THREADS) {
      const int t = c0 + e / 12, k = e % 12;
      const cd2 *raw_t = ce_raw + (t - r0) * 12;
      const bool leftmost = ((sh0 < sh1) ? 1 : 0) ^ (t & 1);
      cd2 total = mk(0, 0);
      int n_total = 0;
      for (int i = k - 1; i <= k + 1; ++i) if (i >= 0 && i <= 11) { total = cadd(total, raw_t[i]); ++n_total; }
      int lo, hi;
      if (sh0 == sh1) { lo = k - 1; hi = k + 1; } else if (leftmost) { lo = k - 1; hi = k; } else { lo = k; hi = k + 1; }
      if (t != 0) {
        cd2 s = mk(0, 0);
        for (int i = lo; i <= hi; ++i) if (i >= 0 && i <= 11) { s = cadd(s, raw_t[i - 12]); ++n_total; }
        total = cadd(total, s);
      }
      if (t != n_rs - 1) {
        cd2 s = mk(0, 0);
        for (int i = lo; i <= hi; ++i) if (i >= 0 && i <= 11) { s = cadd(s, raw_t[i + 12]); ++n_total; }
        total = cadd(total, s);
      }
      ce_filt[e] = cdivr(total, (double)n_total);
    }
    __syncthreads();
    PH(22);
    // noise power (ref :1470): this chunk's share of sum |filtered - raw|^2; the consumers add the
    // chunk partials in chunk order and divide by 12 n_rs (np_from_partials)
    cd2 part = mk(0, 0);
    for (int e = tid; e < (c1 - c0) * 12; e += CE_THREADS) {
      const cd2 d = csub(ce_filt[e], ce_raw[e + (c0 - r0) * 12]);
      part.re += d.re * d.re + d.im * d.im;
    }
    const cd2 tot = block_sum(part, red);
    if (tid == 0) { sc[CS_NPP + port * 8 + chunk] = tot.re; if (chunk == 0) sc[CS_NRS + port] = (double)n_rs; }
    // piecewise-planar interpolation between consecutive RS rows (ref :1237-1351): one thread
    // walks the triangle strip of one row pair exactly as the reference does
    if (chunk == 0) for (int xs = tid; xs < NSC; xs += CE_THREADS) {   // first RS row: plain 1-D interpolation (ref :1250-1252)
      // interp1 (ref dsp.h:151-185: bisection with round_i midpoint, linear, extrapolating) over the
      // edge-extended row, vertices fetched through ext_vertex instead of materialised arrays
      const int n = ext_len(sh0);
      const double x = (double)xs;
      unsigned l = 0, r = (unsigned)n - 1;
      int xm; cd2 vm;
      while (r - l > 1) {
        const unsigned mid = (unsigned)d_round_i((r + l) / 2.0);
        ext_vertex(ce_filt, sh0, (int)mid, xm, vm);
        if (x >= (double)xm) l = mid; else r = mid;
      }
      int xl, xr; cd2 vl, vr;
      ext_vertex(ce_filt, sh0, (int)l, xl, vl);
      ext_vertex(ce_filt, sh0, (int)r, xr, vr);
      const cd2 d = csub(vr, vl);
      st(&out[(size_t)rs_set(0) * NSC + xs], cadd(vl, cdivr(cscale(d, (x - (double)xl)), ((double)xr - (double)xl))));
    }
    PH(23);
    // One WAVE per output row between the chunk's first and last RS row, one LANE per triangle of the row pair's strip
    // (round 3; before: one thread walked the whole strip of its row, ~25 triangles x 72 columns in sequence, and the
    // kernel was the longest of the per-cell chain).  The reference walks the strip triangle by triangle (ref :1262-1351):
    // triangle l = vertices (S_l, S_l+1, S_l+2) of the sequence S that alternates between the two edge-extended RS rows
    // (starting with the row whose second vertex lies further left), its right boundary is the line through S_l+1 and
    // S_l+2, and a column counter x runs on: triangle l emits the columns from where the earlier triangles stopped up to
    // its boundary at this row.  As a scan: stop_l = max over j <= l of (floor(boundary_j(row)) + 1); triangle l emits
    // [stop_(l-1), floor(boundary_l(row))].  The walk ends after the first triangle at which the two tracked rows of the
    // pair (row 1 and the last one) both stand at column 72 exactly (ref :1337-1339), or when either row runs out of
    // vertices.  Every value is the plane through the same three vertices evaluated with the same expressions, so the
    // result is bit-identical to the sequential walk.
    const int y_first = rs_set(c0), y_last = rs_set(min(c1, n_rs - 1));
    const int lane = tid & 63, wv = tid >> 6;
    for (int yy = y_first + 1 + wv; yy <= y_last; yy += CE_THREADS / 64) {
      // the fused chain only ever reads the channel estimate on PBCH rows (`pbch_only`); the last RS row
      // is kept as the source of the edge copy below
      if (pbch_only && !pbch_row(yy, n_symb) && yy != rs_set(n_rs - 1)) continue;
      int t;
      if (port <= 1) {
        const int j = yy / n_symb, rem = yy - j * n_symb;
        t = (rem == 0) ? 2 * j - 1 : (rem <= n_symb - 3 ? 2 * j : 2 * j + 1);
      } else t = (yy - 2) / n_symb;
      const int s_top = (t & 1) ? sh1 : sh0, s_bot = (t & 1) ? sh0 : sh1;
      const cd2 *top = ce_filt + (t - c0) * 12, *bot = ce_filt + (t + 1 - c0) * 12;
      const int n_top = ext_len(s_top), n_bot = ext_len(s_bot);
      const int y_top = rs_set(t), y_bot = rs_set(t + 1);
      const int spacing = y_bot - y_top;
      int x1t, x1b; cd2 dummy;
      ext_vertex(top, s_top, 1, x1t, dummy);
      ext_vertex(bot, s_bot, 1, x1b, dummy);
      const bool top_first = x1t < x1b;
      // S_k: even k from the first row, odd k from the other; the strip ends where either row has no vertex left
      const int n_a = top_first ? n_top : n_bot, n_b = top_first ? n_bot : n_top;      // counts of the first / second row
      const int n_seq = (n_a <= n_b) ? 2 * n_a : 2 * n_b + 1;                          // entries of S: a0 b0 a1 b1 ... while both exist
      const int n_tri = min(n_seq - 2, 64);
      int tx[3], ty[3]; cd2 tv[3];
      const bool have = lane < n_tri;
      if (have) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int k = lane + q;
          const bool from_top = ((k & 1) == 0) == top_first;
          if (from_top) { ext_vertex(top, s_top, k >> 1, tx[q], tv[q]); ty[q] = y_top; }
          else { ext_vertex(bot, s_bot, k >> 1, tx[q], tv[q]); ty[q] = y_bot; }
        }
      } else {
#pragma unroll
        for (int q = 0; q < 3; ++q) { tx[q] = 0; ty[q] = q; tv[q] = mk(0, 0); }
      }
      const double x1 = tx[1], x2 = tx[2], y1 = ty[1], y2 = ty[2];
      const double a_l = (x1 - x2) / (y1 - y2);
      const double b_l = (y1 * x2 - y2 * x1) / (y1 - y2);
      const double bd = a_l * yy + b_l, bd1 = a_l * (y_top + 1) + b_l, bds = a_l * (y_top + spacing) + b_l;
      // where the three column counters stand after this triangle if nothing stood further right before: counter =
      // max(counter, floor(bound) + 1) whenever counter <= bound; inclusive prefix max over the triangles
      int m_mine = have ? (int)floor(bd) + 1 : 0, m_r1 = have ? (int)floor(bd1) + 1 : 0, m_rs = have ? (int)floor(bds) + 1 : 0;
      if (m_mine < 0) m_mine = 0;
      if (m_r1 < 0) m_r1 = 0;
      if (m_rs < 0) m_rs = 0;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {      // whole wave: n_tri may reach 64
        const int o0 = __shfl_up(m_mine, off), o1 = __shfl_up(m_r1, off), o2 = __shfl_up(m_rs, off);
        if (lane >= off) { m_mine = max(m_mine, o0); m_r1 = max(m_r1, o1); m_rs = max(m_rs, o2); }
      }
      int x_from = __shfl_up(m_mine, 1);
      if (lane == 0) x_from = 0;
      // the walk stops after the first triangle that leaves both tracked rows at column 72
      const unsigned long long done = __ballot(have && m_r1 == 72 && m_rs == 72);
      const int l_end = done ? (int)__builtin_ctzll(done) : 63;
      if (have && lane <= l_end && (double)x_from <= bd) {
        // plane through the three vertices (the reference solves the 3x3 system with a LAPACK
        // inverse, ref :1293-1312; same plane, rounding differs at the 1e-14 level)
        const double dx1 = tx[1] - tx[0], dy1 = ty[1] - ty[0], dx2 = tx[2] - tx[0], dy2 = ty[2] - ty[0];
        const double det = dx1 * dy2 - dx2 * dy1;
        const cd2 d1 = csub(tv[1], tv[0]), d2 = csub(tv[2], tv[0]);
        const cd2 a_p = cdivr(csub(cscale(d1, dy2), cscale(d2, dy1)), det);
        const cd2 b_p = cdivr(csub(cscale(d2, dx1), cscale(d1, dx2)), det);
        for (int x = x_from; (double)x <= bd && x <= 71; ++x) {
          const cd2 v = cadd(cadd(tv[0], cscale(a_p, (double)(x - tx[0]))), cscale(b_p, (double)(yy - ty[0])));
          st(&out[(size_t)yy * NSC + x], v);
        }
      }
    }
    __syncthreads();
    PH(24);
    // rows outside the RS span copy the nearest RS row (ref :1356-1361)
    // (the first RS row is written by chunk 0, the last one by the chunk that owns the last row pair)
    const int first = rs_set(0), last = rs_set(n_rs - 1);
    if (chunk == 0)
      for (int e = tid; e < first * NSC; e += CE_THREADS)
        if (!pbch_only || pbch_row(e / NSC, n_symb)) out[e] = out[(size_t)first * NSC + e % NSC];
    if (c1 >= n_rs - 1 && c0 <= max(n_rs - 2, 0))
      for (int e = (last + 1) * NSC + tid; e < n_ofdm * NSC; e += CE_THREADS)
        if (!pbch_only || pbch_row(e / NSC, n_symb)) out[e] = out[(size_t)last * NSC + e % NSC];
    __syncthreads();
  }
}

#undef rs_set

// sigpower(filtered - raw) of one port from k_chan_est's per-chunk partial sums (ref :1470)
__device__ __forceinline__ double np_from_partials(const double *sc, int port) {
  double s = 0;
  for (int q = 0; q < CE_NCHUNK; ++q) s += sc[CS_NPP + port * 8 + q];
  return s / ((double)sc[CS_NRS + port] * 12);
}

// ------------------------------------------------------------------------ PBCH decode
// One WAVE per (cell, candidate): candidate = frame_timing_guess*3 + {1,2,4 ports}.  The equalised symbols stay in
// registers (symbol pairs go straight through the soft demodulator), the 1920 LLRs go through LDS, and the 64
// tail-biting trellises run one per lane with their path metrics in registers (lte_device.h).  Four waves (= one per
// SIMD) share a workgroup: in the pipelined chain a workgroup of this kernel can only start where a resident
// correlation workgroup has retired, and a workgroup should fill the slot it takes (profiles/r03/experiments: the chain's
// cost is the slots it keeps empty, not slower workgroups).
// Round 6: the waves are INDEPENDENT workers on one task list, candidate-major: task k = (candidate k / n, cell k % n); wave w of
// workgroup b takes tasks 4 b + w, + 4 gridDim, ...  Workgroups are dispatched in ascending order, so the list is worked through
// in ascending order as slots free up.  The reference tries the candidates of a cell in order and stops at the first that passes
// (ref :1547, :1567, :1638-1686): a candidate behind a passing one can never be chosen, so a task whose cell already shows a
// passing earlier candidate is skipped -- by the time candidate c of a cell comes up, its candidate c - 1 was dispatched a whole
// sweep over the cells earlier.  (A flag that is not visible yet costs a decode, never a result: k_mib_select takes the first
// passing candidate in order.)  Rounds 4-5 ran three launches of four candidates and skipped whole ranges: 8.0 of 12
// candidates decoded per decodable cell on average, now ~6.5-7, in one launch.
// No survivor words (lte_device.h: two-pass decoder): 15 KB of LLRs + 1 KB per wave, no scratch.
#define PB_THREADS 64                 // lanes of one candidate
#define PB_CANDS 4                    // independent waves per workgroup
// pbch_extract (ref :1503-1520) + equalisation (ref :1571-1612) + soft demodulation + descrambling of one candidate by one wave: the
// LLRs go to e_est (LDS).
static __device__ __forceinline__ void pbch_llr_wave(const lcs_cell &c, const double *__restrict__ sc, const double2 *__restrict__ g, const double2 *__restrict__ cep,
                                                  const uint8_t *__restrict__ scr, int guess, int n_ports, int tid, double *__restrict__ e_est) {
  const int n_symb = cell_n_symb(c), id = cell_id(c);
  const int m_bit = (c.cp_type == LCS_CP_NORMAL) ? 1920 : 1728;
  const int n_sym = m_bit / 2, per_frame = n_sym / 4;
  const int v3 = d_imod(id, 3);
  const int r0 = (v3 == 0) ? 1 : 0, r1 = (v3 == 2) ? 1 : 2;     // the two residues != v3, ascending
  const double np0 = np_from_partials(sc, 0), np1 = np_from_partials(sc, 1), np2 = np_from_partials(sc, 2), np3 = np_from_partials(sc, 3);
  const int start = guess * 10 * 2 * n_symb;
  for (int pr = tid; pr < n_sym / 2; pr += PB_THREADS) {         // one symbol pair per lane and round
    cd2 x[2], ha[2], hb[2], syms[2];
    double npv[2];
    const int t = 2 * pr;
    // the two antenna ports this pair is equalised with (ref :1582-1611): port 0 (and 1) for one / two ports; with four, pairs
    // alternate between ports (0, 2) and (1, 3)
    const int pa = (n_ports == 4 && (t & 3) != 0) ? 1 : 0, pb = (n_ports == 2) ? 1 : (n_ports == 4 ? pa + 2 : 0);
    for (int q = 0; q < 2; ++q) {
      const int idx = t + q;
      const int fr = idx / per_frame;
      int rem = idx % per_frame, sym;
      if (rem < 48) sym = 0; else if (rem < 96) { sym = 1; rem -= 48; } else if (rem < 168) { sym = 2; rem -= 96; } else { sym = 3; rem -= 168; }
      const bool has_rs = (sym == 0) || (sym == 1) || (sym == 3 && n_symb == 6);
      const int scx = has_rs ? (3 * (rem / 2) + ((rem & 1) ? r1 : r0)) : rem;
      const int row = start + fr * 10 * 2 * n_symb + n_symb + sym;
      x[q] = ld(&g[(size_t)row * NSC + scx]);
      ha[q] = ld(&cep[((size_t)pa * ROWS + row) * NSC + scx]);
      hb[q] = ld(&cep[((size_t)pb * ROWS + row) * NSC + scx]);
    }
    if (n_ports == 1) {
      for (int q = 0; q < 2; ++q) {
        const cd2 gain = cconj(cdiv(ha[q], mk(cabs2(ha[q]), 0)));
        syms[q] = cmul(x[q], gain);
        npv[q] = np0 * cabs2(gain);
      }
    } else {
      const cd2 h1 = cdivr(cadd(ha[0], ha[1]), 2), h2 = cdivr(cadd(hb[0], hb[1]), 2);
      const double np_temp = (n_ports == 2) ? (np0 + np1) / 2 : (pa == 0 ? (np0 + np2) / 2 : (np1 + np3) / 2);
      const double scale = h1.re * h1.re + h1.im * h1.im + h2.re * h2.re + h2.im * h2.im;
      const cd2 s0 = cdivr(cadd(cmul(cconj(h1), x[0]), cmul(h2, cconj(x[1]))), scale);
      const cd2 s1 = cconj(cdivr(cadd(cmul(mk(-h2.re, h2.im), x[0]), cmul(h1, cconj(x[1]))), scale));
      const double a1 = hypot(h1.re, h1.im) / scale, a2 = hypot(h2.re, h2.im) / scale;
      const double npp = (a1 * a1 + a2 * a2) * np_temp;
      const double s2 = pow(2.0, 0.5);
      syms[0] = cscale(s0, s2); syms[1] = cscale(s1, s2);
      npv[0] = npp; npv[1] = npp;
    }
    // soft demodulation (exact log-MAP, lte_device.h) and descrambling
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int l = t + q;
      double l0, l1;
      qpsk_llr(syms[q], npv[q], l0, l1);
      if (scr[2 * l]) l0 = -l0;
      if (scr[2 * l + 1]) l1 = -l1;
      e_est[2 * l] = l0; e_est[2 * l + 1] = l1;
    }
  }
}
__global__ __launch_bounds__(PB_THREADS * PB_CANDS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_pbch(const lcs_cell *__restrict__ cells, int *__restrict__ n_work,
                                                      const double2 *__restrict__ tfg_comp, const double2 *__restrict__ ce,
                                                      double *__restrict__ scratch, const uint8_t *__restrict__ pbch_scr,
                                                      const int16_t *__restrict__ derm_inv /*[2][120][16]*/) {
  LCS_TAIL_PRIO();
  __shared__ double llr_all[PB_CANDS][1920];
  __shared__ double d_est_all[PB_CANDS][3][40];
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), tid = threadIdx.x & 63;
  double *e_est = llr_all[wv];
  double (*d_est)[40] = d_est_all[wv];
  const int n_cells = n_work[0], n_tasks = 12 * n_cells;
  for (int task = (int)blockIdx.x * PB_CANDS + wv; task < n_tasks; task += (int)gridDim.x * PB_CANDS) {
    const int cand = task / n_cells, it = task - cand * n_cells;
    const int guess = cand / 3, n_ports = (cand % 3 == 2) ? 4 : (cand % 3) + 1;
    double *sc = scratch + (size_t)it * CS_SIZE;
    bool decided = false;
    for (int k = 0; k < cand; ++k) decided |= __hip_atomic_load(&sc[CS_CAND + k * 4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0.0;
    if (!decided) {                   // (a skipped candidate's flag stays 0, as k_cell_prep left it)
      const lcs_cell c = cells[it];
      const int m_bit = (c.cp_type == LCS_CP_NORMAL) ? 1920 : 1728;
      lcs_wave_sync();                // the previous task's LLRs are no longer read
      PH(0);
      pbch_llr_wave(c, sc, tfg_comp + (size_t)it * ROWS * NSC, ce + ((size_t)it * 4) * ROWS * NSC, pbch_scr + (size_t)cell_id(c) * 1920, guess, n_ports, tid, e_est);
      PH(1);
      int ok = 0;
      unsigned long long bits40 = 0ull;
      pbch_decode_wave(e_est, d_est, derm_inv, m_bit, n_ports, tid, ok, bits40);
      if (tid == 0) {
        const unsigned bits24 = (unsigned)(bits40 & 0xffffffull);
        sc[CS_CAND + cand * 4 + 1] = (double)bits24;
        __hip_atomic_store(&sc[CS_CAND + cand * 4 + 0], (double)ok, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        atomicAdd(&n_work[3], 1);       // statistics only (lcs_last_batch_stats): candidates decoded, i.e. not skipped by the early exit
      }
      PH(4);
    }
  }
}

// first passing candidate in the reference's loop order wins (ref :1547, :1567, :1638-1686); in the fused chain the
// finished record also goes back to its (buffer, peak) place in the peak table (round 2: a kernel of its own)
__global__ __launch_bounds__(64) void k_mib_select(lcs_cell *__restrict__ cells, const int *__restrict__ n_work, const double *__restrict__ scratch,
                                                   lcs_cell *__restrict__ peaks, const WorkItem *__restrict__ items) {
  LCS_TAIL_PRIO();
  const int it = blockIdx.x * blockDim.x + threadIdx.x;
  if (it >= *n_work) return;
  const double *sc = scratch + (size_t)it * CS_SIZE;
  if (sc[CS_OOB] == 0.0) {
    for (int cand = 0; cand < 12; ++cand) {
      if (sc[CS_CAND + cand * 4] == 0.0) continue;
      const unsigned bits = (unsigned)sc[CS_CAND + cand * 4 + 1];
      const int guess = cand / 3, n_ports = (cand % 3 == 2) ? 4 : (cand % 3) + 1;
      lcs_cell c = cells[it];
      auto bit = [&](int i) { return (int)((bits >> i) & 1u); };
      c.n_ports = n_ports;
      const int bw = bit(0) * 4 + bit(1) * 2 + bit(2);
      const int bwt[6] = {6, 15, 25, 50, 75, 100};
      if (bw < 6) c.n_rb_dl = bwt[bw];
      c.phich_duration = bit(3) ? 2 : 1;
      c.phich_resource = 1 + bit(4) * 2 + bit(5);
      const signed char sfn_temp = (signed char)(128 * bit(6) + 64 * bit(7) + 32 * bit(8) + 16 * bit(9) + 8 * bit(10) + 4 * bit(11) + 2 * bit(12) + bit(13));   // quirk Q10
      c.sfn = d_imod((int)sfn_temp * 4 - guess, 1024);
      cells[it] = c;
      break;
    }
  }
  if (peaks) peaks[(size_t)items[it].slot * LCS_MAXP + items[it].peak] = cells[it];
}

// ------------------------------------------------------------------------------ launch
// workgroups loop over the work list; c->grid_items of them per list axis (64: a typical 64-buffer batch in one round)
int lcs_launch_gather_work(lcs_ctx *c, int n_buf, int skip, int limit) {
  hipLaunchKernelGGL(k_gather_work, dim3(1), dim3(64), 0, c->stream, c->peaks, c->npeaks, n_buf, skip, limit > 0 ? limit : std::min(c->max_work, c->percell_cap),
                     c->st_open ? c->st_dtracked : nullptr, c->st_dntracked, c->work_items, c->n_work,
                     c->cells_out);
  HIPCHK(c, hipGetLastError());
  return LCS_OK;
}
int lcs_launch_pack_results(lcs_ctx *c, int n_buf, bool full) {
  int *hdr = reinterpret_cast<int *>(c->res_pack);
  hipLaunchKernelGGL(k_pack_results, dim3(1), dim3(64), 0, c->stream, c->peaks, c->npeaks, n_buf, full ? 1 : 0, full ? c->n_work : nullptr, hdr, hdr + 8,
                     reinterpret_cast<lcs_cell *>(reinterpret_cast<char *>(c->res_pack) + lcs_pack_rec_offset(n_buf)));
  HIPCHK(c, hipGetLastError());
  return LCS_OK;
}
int lcs_launch_tfg(lcs_ctx *c, uint32_t n_cap, bool with_rs) {
  hipLaunchKernelGGL(k_cell_prep, dim3(c->grid_items), dim3(CP_THREADS), 0, c->stream, c->cells_out, c->work_items, c->n_work, c->params,
                     c->d_pn_jump, c->tfg_ts, c->cell_scratch, c->tfg_desc, with_rs ? 3 : 1, c->needed_rows_only ? 1 : 0);
  const CapSrc cs = lcs_cap_src(c, n_cap);
#define TFG_LAUNCH(KIND) hipLaunchKernelGGL(k_tfg<KIND>, dim3(LCS_TFG_GRID), dim3(TFG_THREADS), 0, c->stream, c->work_items, c->n_work, cs, n_cap, \
                                            c->cell_scratch, c->tfg_desc, c->tfg, c->needed_rows_only ? 1 : 0)
  if (cs.c8) TFG_LAUNCH(0);
  else if (cs.c32) TFG_LAUNCH(1);
  else TFG_LAUNCH(2);
#undef TFG_LAUNCH
  HIPCHK(c, hipGetLastError());
  return LCS_OK;
}
int lcs_launch_rs_build(lcs_ctx *c) {
  hipLaunchKernelGGL(k_cell_prep, dim3(c->grid_items), dim3(CP_THREADS), 0, c->stream, c->cells_out, c->work_items, c->n_work, c->params,
                     c->d_pn_jump, c->tfg_ts, c->cell_scratch, c->tfg_desc, 2, 0);
  HIPCHK(c, hipGetLastError());
  return LCS_OK;
}
// apply_grid: also write the corrected grid (the stage entry point); the fused chain leaves the correction to k_chan_est
int lcs_launch_tfoec(lcs_ctx *c, bool apply_grid, int parts) {
  hipLaunchKernelGGL(k_tfoec_est, dim3(c->grid_items, std::min(std::max(parts, 1), TF_PARTS)), dim3(TF_THREADS), 0, c->stream, c->cells_out, c->work_items, c->n_work,
                     c->params, c->tfg, c->tfg_ts, c->cell_scratch, c->tfg_ts_comp);
  if (apply_grid) hipLaunchKernelGGL(k_tfoec_apply, dim3(LCS_TFA_GRID), dim3(TFA_THREADS), 0, c->stream, c->n_work, c->tfg, c->tfg_ts, c->cell_scratch,
                     c->cells_out, c->tfg_comp, c->needed_rows_only ? 1 : 0);
  HIPCHK(c, hipGetLastError());
  return LCS_OK;
}
int lcs_launch_chan_est(lcs_ctx *c) {
  hipLaunchKernelGGL(k_chan_est, dim3(c->grid_items, 4, CE_NCHUNK), dim3(CE_THREADS), 0, c->stream, c->cells_out, c->n_work,
                     c->tfg_comp, (const double2 *)nullptr, (const double *)nullptr, c->cell_scratch, c->ce, c->needed_rows_only ? 1 : 0);
  HIPCHK(c, hipGetLastError());
  return LCS_OK;
}
void lcs_chan_est_np_layout(int *first, int *per_port, int *n_rs_first) { *first = CS_NPP; *per_port = 8; *n_rs_first = CS_NRS; }
// fused: the chain's form -- the grid comes uncorrected from k_tfg (k_chan_est applies k_tfoec_est's corrections to what it
// and k_pbch read), and the finished records go back into the peak table
int lcs_launch_mib(lcs_ctx *c, bool fused) {
  const bool scatter_back = fused;
  hipLaunchKernelGGL(k_chan_est, dim3(c->grid_items, 4, CE_NCHUNK), dim3(CE_THREADS), 0, c->stream, c->cells_out, c->n_work,
                     c->tfg_comp, fused ? (const double2 *)c->tfg : (const double2 *)nullptr, (const double *)c->tfg_ts, c->cell_scratch,
                     c->ce, c->needed_rows_only ? 1 : 0);
  // one launch: independent waves on a candidate-major task list (k_pbch), one task per wave when the work list is as long as the
  // previous batch said (grid_items ~ cells expected + 1/8); a longer list only makes the waves loop
  hipLaunchKernelGGL(k_pbch, dim3(std::max(32, 3 * c->grid_items)), dim3(PB_THREADS * PB_CANDS), 0, c->stream, c->cells_out, c->n_work, c->tfg_comp,
                     c->ce, c->cell_scratch, c->d_pbch_scr, c->d_derm_inv);
  hipLaunchKernelGGL(k_mib_select, dim3((LCS_MAX_WORK + 63) / 64), dim3(64), 0, c->stream, c->cells_out, c->n_work,
                     c->cell_scratch, scatter_back ? c->peaks : nullptr, c->work_items);
  HIPCHK(c, hipGetLastError());
  return LCS_OK;
}
