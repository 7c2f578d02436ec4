// tracker.hip -- LTE-Tracker's per-symbol pipeline for blocks of OFDM symbols of many tracked
// cells at once.
//
// The reference gives every tracked cell a thread that consumes one OFDM symbol at a time (src/tracker_thread.cpp:
// 823-1068): get_fd (:91-174: frequency correction, 128-point DFT, the 72 occupied subcarriers, timing / bulk phase
// compensation), the cell-specific reference symbols (:868-890), filter_ce (:176-201) with the power measurements
// (:906-931), the frequency and timing measurements of do_foe (:203-243) and do_toe_v2 (:245-288), interp2d (:383-477)
// and, on the PBCH symbols of four frames, pbch_extract_rt + the decoder of do_mib_decode (:494-529, 555-705).  None
// of that depends on other symbols beyond a window of three reference symbols, so a block of symbols (whole frames,
// starting at slot 0 symbol 0) of n_cells cells is processed as four launches over (cell, symbol), (cell, port) and
// (cell, frame offset).  What stays sequential -- the running bulk phase, a 600-step scalar recurrence per cell --
// is walked by one lane per cell in the preparation kernel; the slow feedback loops that consume the measurements
// (global frequency offset, frame timing, MIB lock counter) are scalar and live on the host (tracker.py).
// All arithmetic is fp64 like the reference.
#include "lte_device.h"
#include <algorithm>
#include <cstring>
#include <vector>

#define FS_LTE 30720000.0
#define TRK_MEAS 9

// (trk_wrap, trk_wrap_certain_interval: lte_device.h)
__device__ __forceinline__ int trk_n_symb(const lcs_track_cell &c) { return c.cp_type == LCS_CP_NORMAL ? 7 : 6; }
__device__ __forceinline__ double trk_sym_len(int cp_type, int sym) {            // samples from the previous DFT to this one's
  return (cp_type == LCS_CP_EXTENDED) ? 128 + 32 : ((sym == 0) ? 128 + 10 : 128 + 9);
}

// Per cell: RS_DL, the running bulk phase of get_fd at every symbol (:151-153), and per port the list of symbols that
// carry its reference symbols.  128 threads.  The bulk phase is a scalar recurrence with a WRAP per step (one lane walks it,
// as the reference does), but nothing else about it is serial: the per-symbol increments are formed by all lanes into LDS
// first and the results leave through LDS afterwards (round 3 loaded a frequency offset and stored a phase inside every step
// of the walk: 0.29 us per step, the latency of the load -- 282 us of the block's 1.09 ms), and a port's symbol list is one
// frame's pattern repeated (round 3: four lanes walking all symbols with a global read each).  Round 5: a step of the walk was
// still 0.15 us -- the fp64 DIVISION inside WRAP, whose only use is its floor: below.
#define TRK_PREP_CHUNK 1024
__global__ __launch_bounds__(128) void k_trk_prep(lcs_track_cell *__restrict__ cells, int n_sym, const double *__restrict__ freq_off,
                                                 const uint32_t *__restrict__ pn_jump, double *__restrict__ rs /*[c][140][24]*/,
                                                 double *__restrict__ shift /*[c][140][4]*/, double *__restrict__ bpo /*[c][n_sym]*/,
                                                 int *__restrict__ rs_idx /*[c][4][max_rs]*/, int *__restrict__ n_rs /*[c][4]*/, int max_rs) {
  __shared__ double s_inc[TRK_PREP_CHUNK], s_b[TRK_PREP_CHUNK], s_w[TRK_PREP_CHUNK], s_nf[TRK_PREP_CHUNK], s_lo[TRK_PREP_CHUNK], s_hi[TRK_PREP_CHUNK];
  __shared__ double s_part[128], s_start;
  __shared__ double s_sh[140 * 4];
  __shared__ int s_fl[4][44], s_cnt[4];
  const int cell = blockIdx.x, tid = threadIdx.x;
  const lcs_track_cell c = cells[cell];
  const int n_symb = trk_n_symb(c), id = c.n_id_2 + 3 * c.n_id_1;
  double *sh = shift + (size_t)cell * 140 * 4;
  for (int e = tid; e < 140 * 4; e += 128) sh[e] = -1.0;
  __syncthreads();
  if (tid < 60) {
    const int slot = tid / 3, t = tid % 3;
    const int sym = (t == 2) ? (n_symb - 3) : t, row = slot * n_symb + sym;
    rs_dl_row(slot, t, id, c.cp_type, n_symb, pn_jump, rs + ((size_t)cell * 140 + row) * 24, sh + row * 4);
  }
  // the bulk phase, TRK_PREP_CHUNK symbols at a time.  WRAP's quotient floor (how many turns a step folds away) does not need the
  // walk: all lanes form the unwrapped phase by a prefix sum, its turn count after every step and so each step's
  // n * floor -- with the SAME product WRAP forms -- and the interval of k in which that floor is certainly the one WRAP's division
  // gives (1e-9 n inside the quotient's integer bounds; the prefix sum is good to 1e-12).  The walk is then four dependent additions a
  // step on WRAP's own values and checks every k against its interval; a step outside it (never seen) sends the chunk through
  // WRAP as written.
  double b = c.bulk_phase_offset;                     // carried by thread 64
  if (tid == 64) s_start = b;
  const double wn = M_PI - (-M_PI);
  for (int base = 0; base < n_sym; base += TRK_PREP_CHUNK) {
    const int n = min(TRK_PREP_CHUNK, n_sym - base);
    constexpr int PER = TRK_PREP_CHUNK / 128;
    double loc[PER], run = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int e = tid * PER + q, i = base + e;
      const double inc = (e < n) ? 2 * M_PI * trk_sym_len(c.cp_type, i % n_symb) * (1 / (FS_LTE / 16)) * -freq_off[(size_t)cell * n_sym + i] : 0.0;
      if (e < n) s_inc[e] = inc;
      run += inc;
      loc[q] = run;
    }
    s_part[tid] = run;
    __syncthreads();
    double off = 0;
    for (int t = 0; t < tid; ++t) off += s_part[t];
    const double start = s_start;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int e = tid * PER + q;
      if (e < n) s_w[e] = floor((start + (off + loc[q]) + M_PI) / wn);            // turns folded away up to and including step e
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int e = tid * PER + q;
      if (e < n) {
        double nf, lo, hi;
        trk_wrap_certain_interval(s_w[e] - (e ? s_w[e - 1] : 0.0), nf, lo, hi);
        s_nf[e] = nf; s_lo[e] = lo; s_hi[e] = hi;
      }
    }
    __syncthreads();
    if (tid == 64) {
      const double b0 = b;
      int certain = 1;                                  // (no short-circuit: a branch per step would put the loads of a step in sequence)
#pragma unroll 4
      for (int e = 0; e < n; ++e) {
        const double inc = s_inc[e], nf = s_nf[e], lo = s_lo[e], hi = s_hi[e];
        const double k = (b + inc) - (-M_PI);
        certain &= (int)(k >= lo) & (int)(k < hi);
        b = (k - nf) + (-M_PI);
        s_b[e] = b;
      }
      if (!certain) {
        b = b0;
        for (int e = 0; e < n; ++e) { b = trk_wrap(b + s_inc[e], -M_PI, M_PI); s_b[e] = b; }
      }
      s_start = b;
    }
    __syncthreads();
    for (int e = tid; e < n; e += 128) bpo[(size_t)cell * n_sym + base + e] = s_b[e];
    __syncthreads();
  }
  if (tid == 64) cells[cell].bulk_phase_offset = b;
  // per port: the rows of ONE frame that carry its reference symbols, then the list = that pattern frame after frame
  for (int e = tid; e < 140 * 4; e += 128) s_sh[e] = sh[e];
  __syncthreads();
  const int F = 20 * n_symb;
  if (tid < 4) {
    int cnt = 0;
    if (tid < c.n_ports)
      for (int row = 0; row < F; ++row)
        if (s_sh[row * 4 + tid] >= 0.0 && cnt < 44) s_fl[tid][cnt++] = row;
    s_cnt[tid] = cnt;
  }
  __syncthreads();
  for (int port = 0; port < 4; ++port) {
    const int cnt = s_cnt[port];
    int total = 0;
    if (cnt > 0) {
      const int full = n_sym / F, rem = n_sym - full * F;
      int tail = 0;
      for (int k = 0; k < cnt; ++k) tail += (s_fl[port][k] < rem) ? 1 : 0;
      total = min(full * cnt + tail, max_rs);
      for (int m = tid; m < total; m += 128) rs_idx[((size_t)cell * 4 + port) * max_rs + m] = (m / cnt) * F + s_fl[port][m % cnt];
    }
    if (tid == 0) n_rs[cell * 4 + port] = total;
  }
}

// get_fd: one wave per OFDM symbol, 4 symbols per workgroup.
#define TRK_FD_SYM 4
__global__ __launch_bounds__(64 * TRK_FD_SYM) void k_trk_fd(const lcs_track_cell *__restrict__ cells, int n_sym, int sym_first, const double2 *__restrict__ td,
                                                            const double *__restrict__ freq_off, const double *__restrict__ late,
                                                            const double *__restrict__ bpo, double fc_req, double fc_prog, double fs_prog,
                                                            double2 *__restrict__ syms) {
  __shared__ cd2 W[64];
  __shared__ cd2 win[TRK_FD_SYM][128];
  const int cell = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // symbols below sym_first already have their rows in `syms` (continuous tracking: the carried frames of the previous call)
  const int i = sym_first + blockIdx.x * TRK_FD_SYM + wv;
  const bool live = i < n_sym;
  if (tid < 64) { double s, c; sincospi((double)tid / 64.0, &s, &c); W[tid] = mk(c, -s); }
  if (live) {
    const double f_off = freq_off[(size_t)cell * n_sym + i];
    const double k_factor = (fc_req - f_off) / fc_prog;
    const double k = M_PI * (-f_off) / ((fs_prog * k_factor) / 2);                  // fshift_inplace, include/dsp.h:58-69
    const double2 *src = td + ((size_t)cell * n_sym + i) * 128;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int t = lane + 64 * h;
      const double2 x = src[t];
      const cd2 v = cmul(mk(x.x, x.y), cis(k * t));
      win[wv][(t + 126) & 127] = v;                                                 // remove the 2 sample delay (:128-134)
    }
  }
  __syncthreads();
  // 128-point decimation-in-frequency FFT, bit-reversed order out.  A wave owns its window: no workgroup barrier between
  // the stages, and the lane's seven twiddles come out of the table once (their power-of-two strides collide on the banks)
  cd2 twr[7];
#pragma unroll
  for (int stg = 0; stg < 7; ++stg) twr[stg] = W[(lane & ((64 >> stg) - 1)) << stg];
#pragma unroll
  for (int stg = 0; stg < 7; ++stg) {
    const int half = 64 >> stg;
    const int pos = lane & (half - 1);
    const int i0 = ((lane >> (6 - stg)) << (7 - stg)) + pos, i1 = i0 + half;
    cd2 *x = win[wv];
    const cd2 a = x[i0], b = x[i1];
    x[i0] = cadd(a, b);
    x[i1] = cmul(csub(a, b), twr[stg]);
    lcs_wave_sync();
  }
  if (!live) return;
  const lcs_track_cell c = cells[cell];
  (void)c;
  const double kl = 2 * M_PI * late[(size_t)cell * n_sym + i] / 128;
  const double b = bpo[(size_t)cell * n_sym + i];
  const cd2 bpo_coeff = cis(b);
  for (int j = lane; j < 72; j += 64) {
    const int bin = (j < 36) ? 92 + j : j - 35;                                     // :138-141
    cd2 a = cdivr(win[wv][__brev((unsigned)bin) >> 25], sqrt(128.0));
    const int t = (j >= 36) ? j - 35 : 36 - j;
    const double phase = -kl * t;
    cd2 coeff = cis(phase);
    if (j < 36) coeff.im = -coeff.im;
    a = cmul(a, cmul(bpo_coeff, coeff));                                            // :158-165
    st(&syms[((size_t)cell * n_sym + i) * 72 + j], a);
  }
}

// interp72 (:383-400) evaluated at subcarrier t: the segment the reference's running pointers have reached there; slope[a] =
// (filt[a + 1] - filt[a]) / 6, formed once per row (the two divisions were 2/3 of an interpolated element's instructions)
__device__ __forceinline__ cd2 trk_interp72(const cd2 *filt, const cd2 *slope, int shift, int t) {
  int a = (t - shift - 1 >= 0) ? (t - shift - 1) / 6 : 0;
  if (a > 10) a = 10;
  return cadd(cscale(slope[a], (double)(t - (shift + 6 * a))), filt[a]);
}

// Raw channel estimates on the reference symbols, filter_ce + powers + FOE/TOE measurements for every reference symbol that has
// both neighbours, then the 2-D interpolation onto every symbol.
// Round 6: one workgroup per (port, cell, CHUNK of TRK_CE_CH filtered reference symbols) -- rounds 3-5 ran one 512-thread
// workgroup per (port, cell): 256 workgroups on 256 CUs, the filter stage on ~280 of its 512 threads with four 12-tap rows
// (192 registers) per thread and 22 of them spilled, the interpolation 138 sequential rounds of dependent global loads per thread
// (422 us of the block's kernel time).  A chunk stages the raw estimates of its rows (+ 3 rows of halo) in LDS, filters them one
// (row, subcarrier) per thread, takes the per-row measurements from LDS (no arrays in registers, no spills), keeps its filtered
// rows in LDS and interpolates the symbols they bracket from there.  Every value is formed by the same expression in the same
// order as before (the rows are bit-identical; tests/test_tracker.py, test_gpu_configs.py); what a chunk shares with its
// neighbours (the halo rows) it computes again and leaves to its owner to write.
#define TRK_CE_THREADS 256
#define TRK_CE_CH 24          // filtered reference symbols (and interpolation brackets) per workgroup
#define TRK_CE_SYMS 64        // symbols whose brackets are held in LDS at a time
__global__ __launch_bounds__(TRK_CE_THREADS) void k_trk_ce(const lcs_track_cell *__restrict__ cells, int n_sym, const double2 *__restrict__ syms,
                                                          const double *__restrict__ freq_off, const double *__restrict__ frame_timing,
                                                          const double *__restrict__ rs, const double *__restrict__ shift,
                                                          const int *__restrict__ rs_idx, const int *__restrict__ n_rs, int max_rs,
                                                          double fc_req, double fc_prog, double fs_prog, double2 *__restrict__ raw,
                                                          double2 *__restrict__ filt, double *__restrict__ fmeta /*[..][max_rs][4]: tp, sp, sp_raw, np*/,
                                                          double *__restrict__ meas, int *__restrict__ n_meas, double2 *__restrict__ ce,
                                                          double *__restrict__ ce_pw, int *__restrict__ ce_upto) {
  const int port = blockIdx.x, cell = blockIdx.y, chunk = blockIdx.z, tid = threadIdx.x;
  const lcs_track_cell c = cells[cell];
  const size_t cp = (size_t)cell * 4 + port;
  if (port >= c.n_ports) { if (tid == 0 && chunk == 0) { n_meas[cp] = 0; ce_upto[cp] = 0; } return; }
  const int n_symb = trk_n_symb(c), per_frame = 20 * n_symb;
  const int m = n_rs[cp];
  const int nf = (m >= 3) ? m - 2 : 0;
  const int f0 = chunk * TRK_CE_CH;                               // first filtered row (= first bracket) of this chunk
  if (f0 >= nf && chunk > 0) return;                              // (chunk 0 always runs: it owns the raw rows of a port with < 3 reference symbols)
  const int *idx = rs_idx + cp * max_rs;
  const double *sh = shift + (size_t)cell * 140 * 4;
  const double *fo = freq_off + (size_t)cell * n_sym, *ft = frame_timing + (size_t)cell * n_sym;
  double2 *raw_p = raw + cp * max_rs * 12, *filt_p = filt + cp * max_rs * 12;
  double *fm = fmeta + cp * max_rs * 4, *ms = meas + cp * max_rs * TRK_MEAS;
  const bool last = f0 + TRK_CE_CH >= nf;                         // the chunk that owns the tail
  const int nfl = min(TRK_CE_CH + 1, nf - f0);                    // filtered rows computed here: f0 .. f0 + nfl - 1 (one row of halo)
  const int f_own = last ? max(nf - f0, 0) : TRK_CE_CH;           // ... of which it writes the first f_own
  const int r_end = last ? m : min(f0 + nfl + 2, m);              // raw rows staged: f0 .. r_end - 1
  const int r_own = last ? m - f0 : TRK_CE_CH;                    // ... of which it writes the first r_own
  __shared__ cd2 s_raw[(TRK_CE_CH + 3) * 12];
  __shared__ cd2 s_filt[(TRK_CE_CH + 1) * 12];
  __shared__ cd2 s_slope[(TRK_CE_CH + 1) * 12];                   // 11 per row: (filt[a + 1] - filt[a]) / 6
  __shared__ double s_fm[(TRK_CE_CH + 1) * 4];
  __shared__ int s_idx[TRK_CE_CH + 3];
  __shared__ double s_sh[TRK_CE_CH + 3];
  // raw channel estimates (:874-880) of the rows f0 .. r_end - 1 (at most TRK_CE_CH + 3: the filtered rows, their two neighbours and
  // the halo row's)
  for (int e = tid; e < (r_end - f0) * 12; e += TRK_CE_THREADS) {
    const int rl = e / 12, k = e % 12, i = idx[f0 + rl], row = i % per_frame;
    const double shv = sh[row * 4 + port];
    const int sft = d_round_i(shv);
    const cd2 ref = mk(rs[((size_t)cell * 140 + row) * 24 + 2 * k], rs[((size_t)cell * 140 + row) * 24 + 2 * k + 1]);
    const cd2 v = cmul(ld(&syms[((size_t)cell * n_sym + i) * 72 + sft + 6 * k]), cconj(ref));
    s_raw[e] = v;
    if (k == 0) { s_idx[rl] = i; s_sh[rl] = shv; }
    if (rl < r_own) st(&raw_p[(size_t)(f0 + rl) * 12 + k], v);
  }
  __syncthreads();
  // filter_ce :176-201, one (row, subcarrier) per thread: row fl uses the raw rows fl (previous), fl + 1 (current), fl + 2 (next)
  for (int e = tid; e < nfl * 12; e += TRK_CE_THREADS) {
    const int fl = e / 12, t = e % 12;
    const cd2 *P = s_raw + fl * 12, *Cq = P + 12, *N = P + 24;
    const bool up = s_sh[fl] < s_sh[fl + 1];
    cd2 total = mk(0, 0);
    int n_total = 0;
    for (int k = t - 1; k <= t + 1; ++k) if (k >= 0 && k <= 11) { total = cadd(total, Cq[k]); ++n_total; }
    const int lo = up ? t : t - 1;
    cd2 sp_ = mk(0, 0), sn_ = mk(0, 0);
    int n_ind = 0;
    for (int k = lo; k <= lo + 1; ++k) if (k >= 0 && k <= 11) { sp_ = cadd(sp_, P[k]); sn_ = cadd(sn_, N[k]); ++n_ind; }
    total = cadd(total, sp_);
    total = cadd(total, sn_);
    n_total += 2 * n_ind;
    const cd2 F = cdivr(total, (double)n_total);
    s_filt[e] = F;
    if (fl < f_own) st(&filt_p[(size_t)(f0 + fl) * 12 + t], F);
  }
  __syncthreads();
  for (int e = tid; e < nfl * 12; e += TRK_CE_THREADS)
    if (e % 12 < 11) s_slope[e] = cdivr(csub(s_filt[e + 1], s_filt[e]), 6.0);
  // Powers and the FOE / TOE measurements of a row (:908-912, do_foe :203-243, do_toe_v2 :245-288): sixteen lanes per row.  The
  // twelve per-subcarrier terms of every sum are formed in parallel (lane k: subcarrier k) and parked in LDS; each sum is then
  // added up by ONE lane in the reference's order, k = 0 .. 11 from zero -- the same additions, so the same values bit for bit
  // (one thread per row walked ~1500 dependent fp64 instructions: 100 of the kernel's 160 us).
  __shared__ double s_t1[16][12], s_t2[16][12];
  __shared__ cd2 s_c1[16][12], s_c2[16][12], s_c3[16][12];
  __shared__ double s_row[16][4];                                  // np, tp, sp_raw, sp of the row being measured
  __shared__ cd2 s_toe[16][2];
  for (int pass = 0; pass < nfl; pass += 16) {
    const int g = tid >> 4, k = tid & 15, fl = pass + g;          // (a 16-lane group never straddles a wave)
    const bool row_ok = fl < nfl, on = row_ok && k < 12;
    const cd2 *P = s_raw + (row_ok ? fl : 0) * 12, *Cq = P + 12, *N = P + 24, *F = s_filt + (row_ok ? fl : 0) * 12;
    const bool up = row_ok && s_sh[fl] < s_sh[fl + 1];
    const cd2 *A = up ? P : Cq, *B = up ? Cq : P;
    __syncthreads();
    if (on) {
      const cd2 d = csub(Cq[k], F[k]);
      s_t1[g][k] = pow(d.re, 2) + pow(d.im, 2);
      s_t2[g][k] = pow(F[k].re, 2) + pow(F[k].im, 2);
      s_c1[g][k] = cmul(cconj(A[k]), B[k]);
      if (k <= 10) s_c2[g][k] = cmul(cconj(B[k]), A[k + 1]);
      s_c3[g][k] = cmul(cconj(P[k]), N[k]);                       // foe
    }
    __syncthreads();
    if (row_ok && k == 0) {
      double d2 = 0, f2s = 0;
      for (int q = 0; q < 12; ++q) d2 += s_t1[g][q];
      for (int q = 0; q < 12; ++q) f2s += s_t2[g][q];
      const double np = (d2 / 12) * 7 / 6, tp = f2s / 12;         // :908-912
      const double sp_raw = tp - np / 7, sp = (.00001 > sp_raw) ? .00001 : sp_raw;
      s_row[g][0] = np; s_row[g][1] = tp; s_row[g][2] = sp_raw; s_row[g][3] = sp;
      s_fm[fl * 4] = tp; s_fm[fl * 4 + 1] = sp; s_fm[fl * 4 + 2] = sp_raw; s_fm[fl * 4 + 3] = np;
    }
    if (row_ok && k == 1) {
      cd2 toe1 = mk(0, 0);
      for (int q = 0; q < 12; ++q) toe1 = cadd(toe1, s_c1[g][q]);
      s_toe[g][0] = cdivr(toe1, 12);
    }
    if (row_ok && k == 2) {
      cd2 s1 = mk(0, 0), s2 = mk(0, 0);
      for (int q = 0; q <= 4; ++q) s1 = cadd(s1, s_c2[g][q]);
      for (int q = 6; q <= 10; ++q) s2 = cadd(s2, s_c2[g][q]);
      s_toe[g][1] = cdivr(cadd(s1, s2), 10);
    }
    __syncthreads();
    const bool own = row_ok && fl < f_own;                         // the halo row: its owner writes the measurements
    if (on && own) {                                               // the weighted FOE terms need the row's noise power
      const double np = s_row[g][0];
      const double f2 = cabs2(F[k]);
      const double foe_np = np * np + 2 * np * f2;
      const double weight = f2 / foe_np;
      s_c1[g][k] = cscale(s_c3[g][k], weight);
      s_t1[g][k] = foe_np * weight * weight;
      s_t2[g][k] = f2 * weight;
    }
    __syncthreads();
    if (own && k == 0) {
      const double np = s_row[g][0], tp = s_row[g][1], sp_raw = s_row[g][2], sp = s_row[g][3];
      const int ip = s_idx[fl], ic = s_idx[fl + 1], in = s_idx[fl + 2];
      cd2 foe_comb = mk(0, 0);
      double foe_comb_np = 0, wsum = 0;
      for (int q = 0; q < 12; ++q) { foe_comb = cadd(foe_comb, s_c1[g][q]); foe_comb_np += s_t1[g][q]; wsum += s_t2[g][q]; }
      const double scale = 1 / wsum;
      foe_comb = cscale(foe_comb, scale);
      foe_comb_np = foe_comb_np * scale * scale;
      const double frequency_offset = fo[ip];
      const double k_factor = (fc_req - frequency_offset) / fc_prog;
      const double residual_f = atan2(foe_comb.im, foe_comb.re) / (2 * M_PI) /
                                (0.0005 + trk_wrap(ft[in] - ft[ip], -19200.0 / 2, 19200.0 / 2) * (1 / (fs_prog * k_factor)));
      const double residual_f_np = (foe_comb_np / 2 > .001) ? foe_comb_np / 2 : .001;
      const int f = f0 + fl;
      fm[f * 4] = tp; fm[f * 4 + 1] = sp; fm[f * 4 + 2] = sp_raw; fm[f * 4 + 3] = np;
      double *mrow = ms + (size_t)f * TRK_MEAS;
      mrow[0] = ic; mrow[1] = np; mrow[2] = tp; mrow[3] = sp_raw; mrow[4] = sp;
      mrow[5] = frequency_offset + residual_f; mrow[6] = residual_f_np;
    }
    if (own && k == 1) {
      const double np = s_row[g][0], sp = s_row[g][3];
      const cd2 toe1 = cdivr(s_toe[g][0], sqrt(sp)), toe2 = cdivr(s_toe[g][1], sqrt(sp));
      const double delay = -(atan2(toe1.im, toe1.re) + atan2(toe2.im, toe2.re)) / 2 / 3 / (2 * M_PI / 128);
      const double delay_np = (np / sp / 2 / 12 > .001) ? np / sp / 2 / 12 : .001;
      double *mrow = ms + (size_t)(f0 + fl) * TRK_MEAS;
      mrow[7] = ft[s_idx[fl + 1]] + delay; mrow[8] = delay_np;
    }
  }
  __syncthreads();
  // interp2d :402-477: symbol i lies between filtered reference symbols j and j + 1 (idx[j + 1] <= i < idx[j + 2]);
  // symbols in front of the first one repeat its estimate; nothing is produced from the last one on
  const int upto = (nf >= 2) ? idx[nf] : 0;
  if (tid == 0 && chunk == 0) { n_meas[cp] = nf; ce_upto[cp] = upto; }
  if (nf < 2 || f0 > nf - 2) return;                              // no bracket starts in this chunk
  double2 *ce_p = ce + cp * n_sym * 72;
  double *pw_p = ce_pw + cp * n_sym * 4;
  // this chunk's brackets j = f0 .. j_hi - 1 and the symbols they cover; reference symbol j + 1 sits in s_idx[j - f0 + 1]
  const int j_hi = min(f0 + TRK_CE_CH, nf - 1);
  const int i_lo = (chunk == 0) ? 0 : s_idx[1], i_hi = idx[j_hi + 1];
  __shared__ int s_j[TRK_CE_SYMS];
  __shared__ double s_w[TRK_CE_SYMS];
  for (int base = i_lo; base < i_hi; base += TRK_CE_SYMS) {
    const int ns = min(TRK_CE_SYMS, i_hi - base);
    __syncthreads();
    for (int q0 = tid; q0 < ns; q0 += TRK_CE_THREADS) {
      const int i = base + q0;
      int lo = f0, hi = j_hi - 1;                                 // largest j in the chunk with idx[j + 1] <= i (f0 when i is in front of all)
      while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_idx[mid - f0 + 1] <= i) lo = mid; else hi = mid - 1; }
      const int j = lo;
      const int i_prev = s_idx[j - f0 + 1];
      const int sym_prev = i_prev % n_symb;
      double time_diff;
      if (port > 2) time_diff = 0.0005;                           // the reference's `port_num>2`
      else if (c.cp_type == LCS_CP_EXTENDED) time_diff = 3 * (128 + 32) * (1 / (FS_LTE / 16));
      else if (sym_prev == 0) time_diff = 4 * (128 + 9) * (1 / (FS_LTE / 16));
      else time_diff = (2 * (128 + 9) + (128 + 10)) * (1 / (FS_LTE / 16));
      double time_offset = 0;
      for (int q = i_prev; q < i; ++q) {                          // the reference's running sum, same order
        const int sy = q % n_symb;
        if (c.cp_type == LCS_CP_EXTENDED) time_offset += (128 + 32) * (1 / (FS_LTE / 16));
        else if (sy == 6) time_offset += (128 + 10) * (1 / (FS_LTE / 16));
        else time_offset += (128 + 9) * (1 / (FS_LTE / 16));
      }
      s_j[q0] = j - f0;
      s_w[q0] = time_offset / time_diff;
    }
    __syncthreads();
    for (int e = tid; e < ns * 72; e += TRK_CE_THREADS) {
      const int q0 = e / 72, t = e % 72, i = base + q0;
      const int jl = s_j[q0];
      const double w = s_w[q0];
      const int sh_a = (int)s_sh[jl + 1], sh_b = (int)s_sh[jl + 2];
      const cd2 a = trk_interp72(s_filt + jl * 12, s_slope + jl * 12, sh_a, t), b = trk_interp72(s_filt + (jl + 1) * 12, s_slope + (jl + 1) * 12, sh_b, t);
      st(&ce_p[(size_t)i * 72 + t], cadd(a, cscale(csub(b, a), w)));
      if (t < 4) pw_p[i * 4 + t] = s_fm[jl * 4 + t] + (s_fm[(jl + 1) * 4 + t] - s_fm[jl * 4 + t]) * w;
    }
  }
}

// One MIB attempt per (frame offset, cell): pbch_extract_rt (:494-529) + the decoder of do_mib_decode (:555-705).
#define TRK_PB_THREADS 64       // one wave: LLRs through LDS, then the 64 trellises one per lane (lte_device.h)
__global__ __launch_bounds__(TRK_PB_THREADS) void k_trk_mib(const lcs_track_cell *__restrict__ cells, int n_sym, int n_off,
                                                           const double2 *__restrict__ syms, const double2 *__restrict__ ce,
                                                           const double *__restrict__ ce_pw, const int *__restrict__ ce_upto,
                                                           const uint8_t *__restrict__ pbch_scr, const int16_t *__restrict__ derm_inv,
                                                           int *__restrict__ mib_ok, unsigned long long *__restrict__ mib_bits,
                                                           const int *__restrict__ mib_first /* nullable */) {
  __shared__ double e_est[1920];                    // the attempt's LLRs (the decoder keeps no survivor words: lte_device.h)
  __shared__ double d_est[3][40];
  const int off = blockIdx.x, cell = blockIdx.y, tid = threadIdx.x;
  if (mib_first && off < mib_first[cell]) {          // continuous tracking: an earlier call attempted this frame offset already
    if (tid == 0) { mib_ok[(size_t)cell * n_off + off] = -1; mib_bits[(size_t)cell * n_off + off] = 0ull; }
    return;
  }
  const lcs_track_cell c = cells[cell];
  const int n_symb = trk_n_symb(c), per_frame = 20 * n_symb, id = c.n_id_2 + 3 * c.n_id_1;
  const int last = (off + 3) * per_frame + n_symb + 3;            // last PBCH symbol of the attempt
  int upto = n_sym;
  for (int p = 0; p < c.n_ports; ++p) upto = min(upto, ce_upto[cell * 4 + p]);
  if (last >= upto || (c.n_ports != 1 && c.n_ports != 2 && c.n_ports != 4)) {
    if (tid == 0) { mib_ok[(size_t)cell * n_off + off] = -1; mib_bits[(size_t)cell * n_off + off] = 0ull; }
    return;
  }
  const int m_bit = (c.cp_type == LCS_CP_NORMAL) ? 1920 : 1728;
  const int n_syms = m_bit / 2, per_fr = n_syms / 4;
  const int v3 = d_imod(id, 3);
  const int r0 = (v3 == 0) ? 1 : 0, r1 = (v3 == 2) ? 1 : 2;       // the two residues != v3, ascending
  const uint8_t *scr = pbch_scr + (size_t)id * 1920;
  for (int pr = tid; pr < n_syms / 2; pr += TRK_PB_THREADS) {
    cd2 x[2], ha[2], hb[2], sy[2];
    double npa = 0, npb = 0, npv[2];
    const int t = 2 * pr;
    // the two antenna ports this pair is equalised with: port 0 (and 1) for one / two ports; with four, pairs alternate between
    // ports (0, 2) and (1, 3) (the array form h[port][q] indexed by a run-time port lived in scratch memory)
    const int pa = (c.n_ports == 4 && (t & 3) != 0) ? 1 : 0, pb = (c.n_ports == 2) ? 1 : (c.n_ports == 4 ? pa + 2 : 0);
    for (int q = 0; q < 2; ++q) {
      const int ix = t + q, fr = ix / per_fr;
      int rem = ix % per_fr, symn;
      if (rem < 48) symn = 0; else if (rem < 96) { symn = 1; rem -= 48; } else if (rem < 168) { symn = 2; rem -= 96; } else { symn = 3; rem -= 168; }
      const bool has_rs = (symn == 0) || (symn == 1) || (symn == 3 && n_symb == 6);
      const int scx = has_rs ? (3 * (rem / 2) + ((rem & 1) ? r1 : r0)) : rem;
      const int i = (off + fr) * per_frame + n_symb + symn;
      x[q] = ld(&syms[((size_t)cell * n_sym + i) * 72 + scx]);
      ha[q] = ld(&ce[(((size_t)cell * 4 + pa) * n_sym + i) * 72 + scx]);
      hb[q] = ld(&ce[(((size_t)cell * 4 + pb) * n_sym + i) * 72 + scx]);
      if (q == 0) {                                                  // np_pre(port, t): the symbol pair shares an OFDM symbol
        npa = ce_pw[(((size_t)cell * 4 + pa) * n_sym + i) * 4 + 3];
        npb = ce_pw[(((size_t)cell * 4 + pb) * n_sym + i) * 4 + 3];
      }
    }
    if (c.n_ports == 1) {
      for (int q = 0; q < 2; ++q) {
        const cd2 gain = cconj(cdiv(ha[q], mk(cabs2(ha[q]), 0)));
        sy[q] = cmul(x[q], gain);
        npv[q] = npa * cabs2(gain);
      }
    } else {
      const cd2 h1 = cdivr(cadd(ha[0], ha[1]), 2), h2 = cdivr(cadd(hb[0], hb[1]), 2);
      const double np_temp = (npa + npb) / 2;
      const double scale = pow(h1.re, 2) + pow(h1.im, 2) + pow(h2.re, 2) + pow(h2.im, 2);
      const cd2 s0 = cdivr(cadd(cmul(cconj(h1), x[0]), cmul(h2, cconj(x[1]))), scale);
      const cd2 s1 = cconj(cdivr(cadd(cmul(mk(-h2.re, h2.im), x[0]), cmul(h1, cconj(x[1]))), scale));
      const double a1 = hypot(h1.re, h1.im) / scale, a2 = hypot(h2.re, h2.im) / scale;
      const double s2 = pow(2.0, 0.5);
      sy[0] = cscale(s0, s2); sy[1] = cscale(s1, s2);
      npv[0] = npv[1] = (a1 * a1 + a2 * a2) * np_temp;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int l = t + q;
      double l0, l1;
      qpsk_llr(sy[q], npv[q], l0, l1);
      if (scr[2 * l]) l0 = -l0;
      if (scr[2 * l + 1]) l1 = -l1;
      e_est[2 * l] = l0; e_est[2 * l + 1] = l1;
    }
  }
  int ok = 0;
  unsigned long long bits40 = 0;
  pbch_decode_wave(e_est, d_est, derm_inv, m_bit, c.n_ports, tid, ok, bits40);
  if (tid == 0) {
    const int bw[8] = {6, 15, 25, 50, 75, 100, 0, 0};
    const int b0 = (int)(bits40 & 1), b1 = (int)((bits40 >> 1) & 1), b2 = (int)((bits40 >> 2) & 1);
    const int n_rb = bw[b0 * 4 + b1 * 2 + b2];
    const int dur = ((bits40 >> 3) & 1) ? 2 : 1, res = 1 + (int)((bits40 >> 4) & 1) * 2 + (int)((bits40 >> 5) & 1);
    const int fields = (n_rb == c.n_rb_dl) && (dur == c.phich_duration) && (res == c.phich_resource);
    mib_ok[(size_t)cell * n_off + off] = (ok ? 1 : 0) | (fields ? 2 : 0);
    mib_bits[(size_t)cell * n_off + off] = bits40;
  }
}

// ---- display statistics of the tracker thread (do_ac_fd :318-341, do_ac_td :343-371, do_pss_sss_sigpower_ce :754-820)
// from what lcs_track_block left in the workspace: the raw reference-signal estimates, the per-symbol powers and the
// frequency-domain symbols.  The running averages these values feed are scalar recurrences and stay with the caller.
// One workgroup per (port, cell): row f = filtered reference symbol f + 1 of the port (rs_curr of the reference's loop).
__global__ __launch_bounds__(256) void k_trk_acf(const lcs_track_cell *__restrict__ cells, const int *__restrict__ n_meas, int max_rs,
                                                const double2 *__restrict__ raw, const double *__restrict__ meas,
                                                double2 *__restrict__ ac_fd /*[c][4][max_rs][12]*/, double2 *__restrict__ ac_td /*[c][4][max_rs][72]*/) {
  const int port = blockIdx.x, cell = blockIdx.y, tid = threadIdx.x;
  const size_t cp = (size_t)cell * 4 + port;
  if (port >= cells[cell].n_ports) return;
  const int nf = n_meas[cp];
  const double2 *raw_p = raw + cp * max_rs * 12;
  const double *ms = meas + cp * max_rs * TRK_MEAS;
  if (ac_fd)
    for (int e = tid; e < nf * 12; e += 256) {
      const int f = e / 12, d = e % 12;
      const double2 *cur = raw_p + (size_t)(f + 1) * 12;
      cd2 a = mk(0, 0);
      for (int t = 0; t < 12 - d; ++t) a = cadd(a, cmul(cconj(ld(&cur[t])), ld(&cur[t + d])));
      a = cdivr(a, (double)(12 - d));
      st(&ac_fd[(cp * max_rs + f) * 12 + d], cdivr(a, ms[(size_t)f * TRK_MEAS + 4]));
    }
  if (ac_td)
    for (int e = tid; e < nf * 72; e += 256) {
      const int f = e / 72, t = e % 72;
      cd2 o = mk(NAN, NAN);                                       // the 72-deep history is not full before row 71
      if (f >= 71) {
        const double2 *cur = raw_p + (size_t)(f + 1) * 12, *old = raw_p + (size_t)(f + 1 - t) * 12;
        cd2 a = mk(0, 0);
        for (int k = 0; k < 12; ++k) a = cadd(a, cmul(cconj(ld(&cur[k])), ld(&old[k])));
        o = cdivr(cdivr(a, 12.0), ms[(size_t)f * TRK_MEAS + 4]);
      }
      st(&ac_td[(cp * max_rs + f) * 72 + t], o);
    }
}
// One thread per (cell, PSS/SSS pair): the sums run in the reference's order (62-element vectors: nothing to spread).
__global__ __launch_bounds__(64) void k_trk_sync(const lcs_track_cell *__restrict__ cells, int n_cells, int n_sym, int max_hf,
                                                const double2 *__restrict__ syms, const double2 *__restrict__ pss_fd,
                                                const int8_t *__restrict__ sss_fd, double *__restrict__ sync /*[c][max_hf][4]*/,
                                                double2 *__restrict__ sync_ce /*[c][max_hf][72]*/) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_cells * max_hf) return;
  const int cell = e / max_hf, k = e % max_hf;
  const lcs_track_cell c = cells[cell];
  const int n_symb = trk_n_symb(c), slot = (k & 1) * 10;
  const int i_sss = (k >> 1) * 20 * n_symb + slot * n_symb + n_symb - 2;
  if (i_sss + 1 >= n_sym) return;
  const double2 *sss = syms + ((size_t)cell * n_sym + i_sss) * 72, *pss = sss + 72;
  auto pw5 = [](const double2 *v) { double r = 0; for (int t = 0; t < 5; ++t) r += pow(v[t].x, 2) + pow(v[t].y, 2); return r / 5; };
  const double np_blank = (pw5(sss) + pw5(sss + 67) + pw5(pss) + pw5(pss + 67)) / 4;
  const int8_t *sf = sss_fd + (((size_t)c.n_id_1 * 3 + c.n_id_2) * 2 + (slot ? 1 : 0)) * 62;
  const double2 *pf = pss_fd + c.n_id_2 * 62;
  auto ce_sss = [&](int t) { return cmul(ld(&sss[5 + t]), mk((double)sf[t], 0)); };
  auto ce_pss = [&](int t) { return cmul(ld(&pss[5 + t]), cconj(ld(&pf[t]))); };
  double d1 = 0, d2 = 0, tp = 0;
  double2 *oc = sync_ce ? sync_ce + ((size_t)cell * max_hf + k) * 72 : nullptr;
  for (int t = 0; t < 62; ++t) {
    const int lt = (t - 6 > 0) ? t - 6 : 0, rt = (t + 6 < 61) ? t + 6 : 61;
    cd2 a = mk(0, 0), b = mk(0, 0);
    for (int q = lt; q <= rt; ++q) a = cadd(a, ce_sss(q));
    for (int q = lt; q <= rt; ++q) b = cadd(b, ce_pss(q));
    const cd2 sm = cdivr(cadd(a, b), (double)(2 * (rt - lt + 1)));
    const cd2 e1 = csub(sm, ce_sss(t)), e2 = csub(sm, ce_pss(t));
    d1 += pow(e1.re, 2) + pow(e1.im, 2);
    d2 += pow(e2.re, 2) + pow(e2.im, 2);
    tp += pow(sm.re, 2) + pow(sm.im, 2);
    if (oc) st(&oc[5 + t], sm);
  }
  if (oc) for (int t = 0; t < 5; ++t) { st(&oc[t], mk(0, 0)); st(&oc[67 + t], mk(0, 0)); }
  const double np = ((d1 / 62) * 13 / 12 + (d2 / 62) * 13 / 12) / 2;
  tp = tp / 62;
  if (sync) { double *o = sync + ((size_t)cell * max_hf + k) * 4; o[0] = tp; o[1] = tp - np / 13; o[2] = np; o[3] = np_blank; }
}

// ----------------------------------------------------------------------------------- the producer thread's symbol cutter
// LTE-Tracker's producer thread (src/producer_thread.cpp:96-131) stamps every sample of the dongle's stream with a time on the
// cell-independent 1.92 MHz time base -- sample n of a buffer whose first sample has timestamp ts0: WRAP(ts0 + n step, 0, 19200), step =
// (FS_LTE / 16) / (fs_programmed k_factor); a call cuts the symbols k0, k0 + 1, .. of each cell searching from its sample pos0 (a buffer
// cut from its start: 0, 0, 0; a later buffer of the stream: the state the previous call returned) -- and, per tracked cell (:196-246), starts a 128-sample capture at the first sample
// at or after the end of the previous capture whose
//     tdiff = WRAP(timestamp - (frame_timing + target), -9600, 9600)   satisfies   |tdiff| < 0.5  or  0 < tdiff < 3      (:203-213)
// with target = 10 (normal CP) / 32 (extended) for slot 0 symbol 0 and advancing by 137 / 138 / 160 per symbol (:236-241);
// tdiff at the hit travels with the symbol as `late`.  The host cutters (tracker.py cut_symbols, host/TrackCells.cpp) walk the
// samples one by one.  Here: the predicate is a window of ~3.5 samples per symbol period whose position is known in closed form,
// so one thread per (cell, symbol) evaluates the SAME double expressions on the five candidates around the window's start and
// takes the first that passes -- equal to the walk as long as every capture ends before the next window begins, which the kernel
// checks (hit_k >= hit_(k-1) + 128 and the sample before the hit fails the predicate); a cell that violates it (a sample rate far
// off 1.92 MHz) is walked by one thread exactly as the host does (k_trk_cut_walk).  Then one wave per symbol copies its 128 samples.
// (TrkCutCell, trk_cut_cell / _target / _pass / _first / _symbol / _walk: lte_device.h -- __host__ __device__, pinned on the CPU by
// tests/test_track_cut_host.py against the sample-by-sample walk and tracker.py's cutter)
// hit[cell][k], late[cell][k]; flags[cell] |= 1 when the closed form's premise does not hold for the cell
// cells [n_cells][5] = (cp_type, frame_timing, freq_off, first symbol, first sample) as doubles: one host -> device copy per call
#define TRK_CUT_CELL(cell) trk_cut_cell((int)cells[5 * (cell)], cells[5 * (cell) + 1], cells[5 * (cell) + 2], fc_req, fc_prog, fs_prog, ts0, \
                                        (long)cells[5 * (cell) + 3], (long)cells[5 * (cell) + 4])
__global__ __launch_bounds__(256) void k_trk_cut_hits(const double *__restrict__ cells, double fc_req, double fc_prog, double fs_prog, double ts0, uint32_t n_cap,
                                                      int n_sym, int *__restrict__ hit, double *__restrict__ late, int *__restrict__ flags) {
  const int cell = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
  __shared__ long s_h0;
  __shared__ double s_l0;
  const TrkCutCell q = TRK_CUT_CELL(cell);
  if (threadIdx.x == 0) { double l; s_h0 = trk_cut_first(q, n_cap, &l); s_l0 = l; }      // the call's first capture: once per workgroup
  __syncthreads();
  if (k >= n_sym) return;
  const long h0 = s_h0;
  const double l0 = s_l0;
  double lt;
  long h;
  if (!trk_cut_symbol(q, n_cap, k, h0, l0, &h, &lt)) atomicOr(&flags[cell], 1);
  hit[(size_t)cell * n_sym + k] = (int)h;
  late[(size_t)cell * n_sym + k] = lt;
}
// The sample-by-sample walk of the host cutters for the cells k_trk_cut_hits flagged (none at any sample rate a dongle produces),
// and the count of symbols found: one WAVE per cell (lane 0 walks; the count and the clean-up behind it run over all lanes -- one
// thread per cell scanning its 980 entries took 160 us).
__global__ __launch_bounds__(64) void k_trk_cut_walk(const double *__restrict__ cells, double fc_req, double fc_prog, double fs_prog, double ts0, uint32_t n_cap,
                                                     int n_sym, int n_cells, int *__restrict__ hit, double *__restrict__ late, int *__restrict__ flags,
                                                     int *__restrict__ n_cut, long long *__restrict__ pos_next) {
  const int cell = blockIdx.x, lane = threadIdx.x;
  int *h = hit + (size_t)cell * n_sym;
  double *lt = late + (size_t)cell * n_sym;
  if (flags[cell] & 1) {                                   // (uniform over the wave)
    if (lane == 0) {
      (void)trk_cut_walk(TRK_CUT_CELL(cell), n_cap, n_sym, h, lt);
      flags[cell] = 0;                                     // the flags start at zero for the next call: no memset per call
    }
    __threadfence_block();
    __builtin_amdgcn_s_barrier();
  }
  // the first symbol that was not found / does not fit; nothing behind it counts
  int n = n_sym;
  for (int k0 = 0; k0 < n_sym && n == n_sym; k0 += 64) {
    const int k = k0 + lane;
    const unsigned long long miss = __ballot(k < n_sym && h[k] < 0);
    if (miss) n = k0 + (int)__builtin_ctzll(miss);
  }
  for (int k = n + lane; k < n_sym; k += 64) { h[k] = -1; lt[k] = 0.0; }
  if (lane == 0) {
    n_cut[cell] = n;
    pos_next[cell] = n > 0 ? (long long)h[n - 1] + 128 : (long long)cells[5 * cell + 4];      // where the next call's search starts (this buffer's indices)
  }
}
// one wave per (symbol, cell): 128 samples -> complex<double>; FMT 1: the dongle's bytes, (u8 - 127) / 128 (src/producer_thread.cpp:121-124)
template <int FMT>
__global__ __launch_bounds__(256) void k_trk_cut_copy(const void *__restrict__ cap, int n_sym, const int *__restrict__ hit, double2 *__restrict__ td) {
  const int cell = blockIdx.y, k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (k >= n_sym) return;
  const int h = hit[(size_t)cell * n_sym + k];
  double2 *out = td + ((size_t)cell * n_sym + k) * 128;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int m = lane + 64 * u;
    double2 v = make_double2(0.0, 0.0);
    if (h >= 0) {
      if (FMT == LCS_FMT_IQ_U8) {
        const uchar2 b = reinterpret_cast<const uchar2 *>(cap)[(size_t)h + m];
        v = make_double2(((double)b.x - 127.0) / 128.0, ((double)b.y - 127.0) / 128.0);
      } else if (FMT == LCS_FMT_C64) {
        const float2 f = reinterpret_cast<const float2 *>(cap)[(size_t)h + m];
        v = make_double2((double)f.x, (double)f.y);
      } else v = reinterpret_cast<const double2 *>(cap)[(size_t)h + m];
    }
    out[m] = v;
  }
}

// ------------------------------------------------------------------------------------------------------------ host
namespace {
// Where lcs_track_block leaves its intermediate results in the block workspace (lcs_track_stats reads them back)
struct TrkLayout {
  int rs_cap, n_off;
  size_t N, C4;
  double *d_fo, *d_ft, *d_late, *d_bpo, *d_rs, *d_shift, *d_fm, *d_meas;
  int *d_idx, *d_nrs, *d_nmeas, *d_upto, *d_mibok, *d_mibfirst;
  double2 *d_raw, *d_filt;
  unsigned long long *d_mibbits;
  TrkLayout(const lcs_ctx *c, int n_cells, int n_sym) {
    rs_cap = n_sym / 3 + 4;                               // reference symbols of one port in the block: at most 2 per slot of >= 6 symbols
    n_off = std::max(0, n_sym / 120 - 3);                 // frame offsets that could hold four frames (120 = extended-CP frame)
    N = (size_t)n_cells * n_sym; C4 = (size_t)n_cells * 4;
    d_fo = c->trk_meta; d_ft = d_fo + N; d_late = d_ft + N; d_bpo = d_late + N;
    d_rs = c->trk_rs; d_shift = d_rs + (size_t)n_cells * 140 * 24;
    d_idx = c->trk_idx; d_nrs = d_idx + C4 * rs_cap;
    d_raw = c->trk_raw; d_filt = d_raw + C4 * rs_cap * 12;
    d_fm = c->trk_fmeta; d_meas = d_fm + C4 * rs_cap * 4;
    d_nmeas = c->trk_small; d_upto = d_nmeas + C4; d_mibok = d_upto + C4;
    d_mibbits = reinterpret_cast<unsigned long long *>(d_mibok + (((size_t)n_cells * n_off + 1) & ~(size_t)1));
    d_mibfirst = c->trk_small + C4 * 2 + (size_t)n_cells * (n_off + 1) * 3;
  }
};
template <typename T>
int trk_alloc(lcs_ctx *c, T **p, size_t n) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  HIPCHK(c, hipMalloc((void **)p, n * sizeof(T)));
  return LCS_OK;
}
}  // namespace

namespace {
// What the block workspace already holds when lcs_track_stream_block calls in (continuous tracking)
struct TrkCarried {
  int n_tail;              // the first n_tail symbols of every cell have their frequency-domain rows in trk_syms (restored from the previous call)
  const int *mib_first;    // host [n_cells]: frame offsets below were attempted by an earlier call and are not decoded again
};

// The workspace grows to the largest (cells, symbols) shape seen and serves every smaller one: TrkLayout places the arrays
// by the block's own shape, every array's size is monotone in both.  (Round 3 reallocated all eleven buffers -- eleven
// device-wide synchronisations -- whenever the shape changed, which the continuous form does with every call whose block is
// not a whole number of frames.)
int trk_ensure_ws(lcs_ctx *c, int n_cells, int n_sym) {
  c->trk_last_cells = c->trk_last_sym = 0;
  if (n_cells > c->trk_cells_cap || n_sym > c->trk_sym_cap) {
    const int cc = std::max(n_cells, c->trk_cells_cap), cs = std::max(n_sym + n_sym / 8, c->trk_sym_cap);     // head room for the tail's varying length
    const int rs_cap = cs / 3 + 4, n_off = std::max(0, cs / 120 - 3);     // as TrkLayout
    const size_t N = (size_t)cc * cs, C4 = (size_t)cc * 4;
    int rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // the caps describe a complete workspace or nothing: if an allocation below fails, later calls must not take the
    // half-replaced buffers for the old shape
    c->trk_cells_cap = c->trk_sym_cap = 0;
    if ((rc = trk_alloc(c, &c->trk_td, N * 128)) || (rc = trk_alloc(c, &c->trk_meta, N * 4)) || (rc = trk_alloc(c, &c->trk_cells, (size_t)cc)) ||
        (rc = trk_alloc(c, &c->trk_syms, N * 72)) || (rc = trk_alloc(c, &c->trk_rs, (size_t)cc * 140 * 28)) ||
        (rc = trk_alloc(c, &c->trk_idx, C4 * (rs_cap + 1))) || (rc = trk_alloc(c, &c->trk_raw, C4 * rs_cap * 24)) ||
        (rc = trk_alloc(c, &c->trk_fmeta, C4 * rs_cap * (4 + TRK_MEAS))) || (rc = trk_alloc(c, &c->trk_ce, C4 * cs * 72)) ||
        (rc = trk_alloc(c, &c->trk_pw, C4 * cs * 4)) || (rc = trk_alloc(c, &c->trk_small, C4 * 2 + (size_t)cc * (n_off + 1) * 3 + cc)))
      return rc;
    c->trk_cells_cap = cc; c->trk_sym_cap = cs;
  }
  c->trk_last_cells = n_cells; c->trk_last_sym = n_sym;
  return LCS_OK;
}

// td: [n_cells][n_sym][128] on the host (td_on_device 0), on the device (1), or -- td_on_device 2, the continuous form --
// already in trk_td (rows carry->n_tail .. n_sym - 1; the rows before are never read)
int trk_block(lcs_ctx *c, lcs_track_cell *cells, int n_cells, int n_sym, const void *td, int td_on_device,
              const double *freq_off, const double *frame_timing, const double *late, double fc_requested,
              double fc_programmed, double fs_programmed, double *syms, double *ce, double *ce_pw,
              int32_t *ce_upto, double *meas, int max_rs, int32_t *n_meas, int32_t *mib_ok, uint64_t *mib_bits,
              int max_off, float *gpu_ms, const TrkCarried *carry) {
  const int rs_cap = n_sym / 3 + 4, n_off = std::max(0, n_sym / 120 - 3);     // as TrkLayout
  int rc;
  if ((rc = trk_ensure_ws(c, n_cells, n_sym))) return rc;
  const TrkLayout L(c, n_cells, n_sym);
  const size_t N = L.N, C4 = L.C4;
  double *d_fo = L.d_fo, *d_ft = L.d_ft, *d_late = L.d_late, *d_bpo = L.d_bpo, *d_rs = L.d_rs, *d_shift = L.d_shift;
  int *d_idx = L.d_idx, *d_nrs = L.d_nrs, *d_nmeas = L.d_nmeas, *d_upto = L.d_upto, *d_mibok = L.d_mibok;
  double2 *d_raw = L.d_raw, *d_filt = L.d_filt;
  double *d_fm = L.d_fm, *d_meas = L.d_meas;
  unsigned long long *d_mibbits = L.d_mibbits;
  const double2 *d_td = (td_on_device == 1) ? (const double2 *)td : c->trk_td;
  const int sym_first = carry ? carry->n_tail : 0;
  // One reusable host block for the small traffic of a call (round 4): the three metadata arrays go up as ONE copy (d_fo,
  // d_ft, d_late are adjacent in trk_meta), the measurement tables come down into the same block -- no per-call vectors
  // (6 MB zero-filled and 71 k row copies per call before).  Plain pageable memory on purpose: with a PAGE-LOCKED block the
  // copies become asynchronous stream operations on the copy engines, and blocks of different contexts stopped overlapping
  // (three contexts in flight: 43 M symbols/s page-locked, 73 M pageable; profiles/r04/experiments/tracker_staging.txt).
  const size_t n_meas_d = C4 * rs_cap * TRK_MEAS, n_small = C4 * 2 + (size_t)n_cells * n_off, n_bits = (size_t)n_cells * n_off + 1;
  const size_t up_bytes = sizeof(double) * 3 * N + sizeof(lcs_track_cell) * n_cells;
  const size_t down_bytes = sizeof(double) * n_meas_d + sizeof(unsigned long long) * n_bits + sizeof(int) * n_small + sizeof(lcs_track_cell) * n_cells;
  if (up_bytes + down_bytes + 64 > c->trk_hpin_bytes) {
    if (c->trk_hpin) { HIPCHK(c, hipStreamSynchronize(c->stream)); free(c->trk_hpin); c->trk_hpin = nullptr; c->trk_hpin_bytes = 0; }
    c->trk_hpin = malloc(up_bytes + down_bytes + 64);
    if (!c->trk_hpin) { c->err = "out of host memory"; return LCS_ERR_HIP; }
    c->trk_hpin_bytes = up_bytes + down_bytes + 64;
  }
  double *h_up = static_cast<double *>(c->trk_hpin);
  lcs_track_cell *h_cells_up = reinterpret_cast<lcs_track_cell *>(h_up + 3 * N);
  double *h_meas = reinterpret_cast<double *>(static_cast<char *>(c->trk_hpin) + ((up_bytes + 15) & ~(size_t)15));
  unsigned long long *h_bits = reinterpret_cast<unsigned long long *>(h_meas + n_meas_d);
  int *h_small = reinterpret_cast<int *>(h_bits + n_bits);
  lcs_track_cell *h_cells_down = reinterpret_cast<lcs_track_cell *>(h_small + ((n_small + 1) & ~(size_t)1));
  std::memcpy(h_up, freq_off, sizeof(double) * N);
  std::memcpy(h_up + N, frame_timing, sizeof(double) * N);
  std::memcpy(h_up + 2 * N, late, sizeof(double) * N);
  std::memcpy(h_cells_up, cells, sizeof(lcs_track_cell) * n_cells);
  if (!td_on_device) HIPCHK(c, hipMemcpyAsync(c->trk_td, td, sizeof(double2) * N * 128, hipMemcpyHostToDevice, c->stream));
  if (carry && carry->mib_first) HIPCHK(c, hipMemcpyAsync(L.d_mibfirst, carry->mib_first, sizeof(int) * n_cells, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_fo, h_up, sizeof(double) * 3 * N, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->trk_cells, h_cells_up, sizeof(lcs_track_cell) * n_cells, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipEventRecord(c->ev_xc0, c->stream));
  hipLaunchKernelGGL(k_trk_prep, dim3(n_cells), dim3(128), 0, c->stream, c->trk_cells, n_sym, d_fo, c->d_pn_jump, d_rs, d_shift, d_bpo, d_idx,
                     d_nrs, rs_cap);
  if (n_sym > sym_first)
    hipLaunchKernelGGL(k_trk_fd, dim3((n_sym - sym_first + TRK_FD_SYM - 1) / TRK_FD_SYM, n_cells), dim3(64 * TRK_FD_SYM), 0, c->stream, c->trk_cells, n_sym,
                       sym_first, d_td, d_fo, d_late, d_bpo, fc_requested, fc_programmed, fs_programmed, c->trk_syms);
  hipLaunchKernelGGL(k_trk_ce, dim3(4, n_cells, std::max(1, (rs_cap - 2 + TRK_CE_CH - 1) / TRK_CE_CH)), dim3(TRK_CE_THREADS), 0, c->stream, c->trk_cells, n_sym, c->trk_syms, d_fo, d_ft, d_rs, d_shift,
                     d_idx, d_nrs, rs_cap, fc_requested, fc_programmed, fs_programmed, d_raw, d_filt, d_fm, d_meas, d_nmeas, c->trk_ce,
                     c->trk_pw, d_upto);
  if (n_off > 0)
    hipLaunchKernelGGL(k_trk_mib, dim3(n_off, n_cells), dim3(TRK_PB_THREADS), 0, c->stream, c->trk_cells, n_sym, n_off, c->trk_syms, c->trk_ce,
                       c->trk_pw, d_upto, c->d_pbch_scr, c->d_derm_inv, d_mibok, d_mibbits,
                       (carry && carry->mib_first) ? (const int *)L.d_mibfirst : (const int *)nullptr);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(c->ev_xc1, c->stream));
  // results
  HIPCHK(c, hipMemcpyAsync(h_cells_down, c->trk_cells, sizeof(lcs_track_cell) * n_cells, hipMemcpyDeviceToHost, c->stream));
  if (syms) HIPCHK(c, hipMemcpyAsync(syms, c->trk_syms, sizeof(double2) * N * 72, hipMemcpyDeviceToHost, c->stream));
  if (ce) HIPCHK(c, hipMemcpyAsync(ce, c->trk_ce, sizeof(double2) * C4 * n_sym * 72, hipMemcpyDeviceToHost, c->stream));
  if (ce_pw) HIPCHK(c, hipMemcpyAsync(ce_pw, c->trk_pw, sizeof(double) * C4 * n_sym * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(h_meas, d_meas, sizeof(double) * n_meas_d, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(h_small, c->trk_small, sizeof(int) * n_small, hipMemcpyDeviceToHost, c->stream));
  if (n_off > 0) HIPCHK(c, hipMemcpyAsync(h_bits, d_mibbits, sizeof(unsigned long long) * n_cells * n_off, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::memcpy(cells, h_cells_down, sizeof(lcs_track_cell) * n_cells);
  if (gpu_ms) HIPCHK(c, hipEventElapsedTime(gpu_ms, c->ev_xc0, c->ev_xc1));
  rc = LCS_OK;
  for (size_t q = 0; q < C4; ++q) {
    int nm = h_small[q];
    if (n_meas) n_meas[q] = std::min(nm, max_rs);
    if (nm > max_rs) rc = LCS_ERR_OVERFLOW;
    if (ce_upto) ce_upto[q] = h_small[C4 + q];
    if (meas && nm > 0) std::memcpy(meas + q * (size_t)max_rs * TRK_MEAS, h_meas + q * (size_t)rs_cap * TRK_MEAS, sizeof(double) * TRK_MEAS * std::min(nm, max_rs));
  }
  for (int i = 0; i < n_cells; ++i)
    for (int o = 0; o < max_off; ++o) {
      if (mib_ok) mib_ok[(size_t)i * max_off + o] = (o < n_off) ? h_small[C4 * 2 + (size_t)i * n_off + o] : -1;
      if (mib_bits) mib_bits[(size_t)i * max_off + o] = (o < n_off) ? h_bits[(size_t)i * n_off + o] : 0ull;
    }
  if (rc) c->err = "more reference symbols per port than max_rs rows";
  return rc;
}
}  // namespace

extern "C" int lcs_track_block(lcs_ctx *c, lcs_track_cell *cells, int n_cells, int n_sym, const void *td, int td_on_device,
                               const double *freq_off, const double *frame_timing, const double *late, double fc_requested,
                               double fc_programmed, double fs_programmed, double *syms, double *ce, double *ce_pw,
                               int32_t *ce_upto, double *meas, int max_rs, int32_t *n_meas, int32_t *mib_ok, uint64_t *mib_bits,
                               int max_off, float *gpu_ms) {
  if (!c) return LCS_ERR_BAD_ARG;
  if (!cells || !td || !freq_off || !frame_timing || !late || n_cells < 1 || n_sym < 1 || max_rs < 0 || max_off < 0) { c->err = "bad argument"; return LCS_ERR_BAD_ARG; }
  for (int i = 0; i < n_cells; ++i) {
    const lcs_track_cell &t = cells[i];
    if (t.n_id_1 < 0 || t.n_id_1 > 167 || t.n_id_2 < 0 || t.n_id_2 > 2 || (t.cp_type != LCS_CP_NORMAL && t.cp_type != LCS_CP_EXTENDED) ||
        t.n_ports < 1 || t.n_ports > 4) { c->err = "tracked cell needs n_id_1, n_id_2, a known cp_type and 1..4 ports"; return LCS_ERR_BAD_ARG; }
  }
  HIPCHK(c, hipSetDevice(c->device));
  return trk_block(c, cells, n_cells, n_sym, td, td_on_device ? 1 : 0, freq_off, frame_timing, late, fc_requested, fc_programmed, fs_programmed, syms, ce,
                   ce_pw, ce_upto, meas, max_rs, n_meas, mib_ok, mib_bits, max_off, gpu_ms, nullptr);
}

// The producer thread's symbol extraction (see k_trk_cut_hits): d_capbuf and d_td are DEVICE memory.
extern "C" int lcs_track_cut(lcs_ctx *c, const void *d_capbuf, int fmt, uint32_t n_cap, double ts_first, int n_cells, const int32_t *cp_type,
                             const double *frame_timing, const double *freq_off, const int64_t *sym_first, const int64_t *pos_first,
                             double fc_requested, double fc_programmed, double fs_programmed, int n_sym, void *d_td, double *late,
                             int32_t *n_cut, int64_t *pos_next) {
  if (!c) return LCS_ERR_BAD_ARG;
  if (!d_capbuf || !cp_type || !frame_timing || !freq_off || !d_td || !n_cut || n_cells < 1 || n_sym < 1 || n_cap < 128 || !std::isfinite(ts_first) ||
      (fmt != LCS_FMT_C64 && fmt != LCS_FMT_IQ_U8 && fmt != LCS_FMT_C128)) { c->err = "bad argument"; return LCS_ERR_BAD_ARG; }
  for (int i = 0; i < n_cells; ++i) {
    if (cp_type[i] != LCS_CP_NORMAL && cp_type[i] != LCS_CP_EXTENDED) { c->err = "the cutter needs a known cp_type per cell"; return LCS_ERR_BAD_ARG; }
    const double kf = (fc_requested - freq_off[i]) / fc_programmed;
    if (!(fs_programmed * kf > 0) || !std::isfinite(frame_timing[i])) { c->err = "the cutter needs a positive sample rate and a finite frame_timing"; return LCS_ERR_BAD_ARG; }
    if ((sym_first && (sym_first[i] < 0 || sym_first[i] > (int64_t)1 << 40)) || (pos_first && (pos_first[i] < 0 || pos_first[i] > (int64_t)n_cap))) {
      c->err = "sym_first / pos_first out of range (pos_first counts samples of THIS buffer)"; return LCS_ERR_BAD_ARG;
    }
  }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t N = (size_t)n_cells * n_sym;
  if (N > c->trk_cut_cap || n_cells > c->trk_cut_cells_cap) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const size_t capN = std::max(N, c->trk_cut_cap);
    const int capC = std::max(n_cells, c->trk_cut_cells_cap);
    c->trk_cut_cap = 0; c->trk_cut_cells_cap = 0;
    int rc;
    if ((rc = trk_alloc(c, &c->trk_cut_hit, capN + 4 * (size_t)capC + 2)) || (rc = trk_alloc(c, &c->trk_cut_meta, capN + 5 * (size_t)capC))) return rc;
    HIPCHK(c, hipMemsetAsync(c->trk_cut_hit, 0, sizeof(int) * (capN + 4 * (size_t)capC + 2), c->stream));      // the per-cell flags start at zero
    c->trk_cut_cap = capN; c->trk_cut_cells_cap = capC;
  }
  // hit [cells][symbols], then the per-cell counts, flags and next positions at FIXED offsets (the flags keep their zeros from call to call)
  int *d_hit = c->trk_cut_hit, *d_ncut = d_hit + c->trk_cut_cap, *d_flags = d_ncut + c->trk_cut_cells_cap;
  long long *d_pos = reinterpret_cast<long long *>(d_hit + ((c->trk_cut_cap + 2 * (size_t)c->trk_cut_cells_cap + 1) & ~(size_t)1));
  double *d_late = c->trk_cut_meta, *d_cells = d_late + c->trk_cut_cap;
  std::vector<double> h_cells((size_t)5 * n_cells);
  for (int i = 0; i < n_cells; ++i) {
    h_cells[5 * i] = (double)cp_type[i]; h_cells[5 * i + 1] = frame_timing[i]; h_cells[5 * i + 2] = freq_off[i];
    h_cells[5 * i + 3] = sym_first ? (double)sym_first[i] : 0.0; h_cells[5 * i + 4] = pos_first ? (double)pos_first[i] : 0.0;
  }
  HIPCHK(c, hipMemcpyAsync(d_cells, h_cells.data(), sizeof(double) * 5 * n_cells, hipMemcpyHostToDevice, c->stream));      // (pageable: staged before the call returns)
  hipLaunchKernelGGL(k_trk_cut_hits, dim3((n_sym + 255) / 256, n_cells), dim3(256), 0, c->stream, d_cells, fc_requested, fc_programmed,
                     fs_programmed, ts_first, n_cap, n_sym, d_hit, d_late, d_flags);
  hipLaunchKernelGGL(k_trk_cut_walk, dim3(n_cells), dim3(64), 0, c->stream, d_cells, fc_requested, fc_programmed,
                     fs_programmed, ts_first, n_cap, n_sym, n_cells, d_hit, d_late, d_flags, d_ncut, d_pos);
  const dim3 grid((n_sym + 3) / 4, n_cells);
  if (fmt == LCS_FMT_IQ_U8) hipLaunchKernelGGL(k_trk_cut_copy<LCS_FMT_IQ_U8>, grid, dim3(256), 0, c->stream, d_capbuf, n_sym, d_hit, (double2 *)d_td);
  else if (fmt == LCS_FMT_C64) hipLaunchKernelGGL(k_trk_cut_copy<LCS_FMT_C64>, grid, dim3(256), 0, c->stream, d_capbuf, n_sym, d_hit, (double2 *)d_td);
  else hipLaunchKernelGGL(k_trk_cut_copy<LCS_FMT_C128>, grid, dim3(256), 0, c->stream, d_capbuf, n_sym, d_hit, (double2 *)d_td);
  HIPCHK(c, hipGetLastError());
  if (late) HIPCHK(c, hipMemcpyAsync(late, d_late, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(n_cut, d_ncut, sizeof(int) * n_cells, hipMemcpyDeviceToHost, c->stream));
  if (pos_next) HIPCHK(c, hipMemcpyAsync(pos_next, d_pos, sizeof(long long) * n_cells, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LCS_OK;
}

// Statistics of the block the last lcs_track_block call on this context processed (its workspace is read, not recomputed).
extern "C" int lcs_track_stats(lcs_ctx *c, int n_cells, int n_sym, double *ac_fd, double *ac_td, int max_rs, double *sync, double *sync_ce,
                               int max_hf, int32_t *n_hf) {
  if (!c) return LCS_ERR_BAD_ARG;
  if (n_cells < 1 || n_cells != c->trk_last_cells || n_sym != c->trk_last_sym || max_rs < 0 || max_hf < 0) {
    c->err = "lcs_track_stats describes the block of the last lcs_track_block call on this context (same n_cells, n_sym)";
    return LCS_ERR_BAD_ARG;
  }
  HIPCHK(c, hipSetDevice(c->device));
  const TrkLayout L(c, n_cells, n_sym);
  const int rs_cap = L.rs_cap;
  const size_t C4 = L.C4;
  const int hf_cap = n_sym / 60 + 2;                       // PSS/SSS pairs of the block: two per frame of >= 120 symbols
  int rc;
  if (n_cells > c->trk_stat_cells || n_sym > c->trk_stat_sym) {      // grow only, like the block workspace
    const int cc = std::max(n_cells, c->trk_stat_cells), cs = std::max(n_sym + n_sym / 8, c->trk_stat_sym);
    const size_t C4c = (size_t)cc * 4, rsc = (size_t)cs / 3 + 4, hfc = (size_t)cs / 60 + 2;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->trk_stat_cells = c->trk_stat_sym = 0;
    if ((rc = trk_alloc(c, &c->trk_acfd, C4c * rsc * 12)) || (rc = trk_alloc(c, &c->trk_actd, C4c * rsc * 72)) ||
        (rc = trk_alloc(c, &c->trk_sync, (size_t)cc * hfc * 4)) || (rc = trk_alloc(c, &c->trk_syncce, (size_t)cc * hfc * 72)))
      return rc;
    c->trk_stat_cells = cc; c->trk_stat_sym = cs;
  }
  const double2 *d_raw = L.d_raw;
  const double *d_meas = L.d_meas;
  const int *d_nmeas = L.d_nmeas;
  std::vector<int> h_nmeas(C4);
  std::vector<lcs_track_cell> h_cells(n_cells);
  if (ac_fd || ac_td)
    hipLaunchKernelGGL(k_trk_acf, dim3(4, n_cells), dim3(256), 0, c->stream, c->trk_cells, d_nmeas, rs_cap, d_raw, d_meas,
                       ac_fd ? c->trk_acfd : nullptr, ac_td ? c->trk_actd : nullptr);
  hipLaunchKernelGGL(k_trk_sync, dim3((n_cells * hf_cap + 63) / 64), dim3(64), 0, c->stream, c->trk_cells, n_cells, n_sym, hf_cap, c->trk_syms,
                     c->d_pss_fd, c->d_sss_fd, c->trk_sync, c->trk_syncce);
  HIPCHK(c, hipGetLastError());
  std::vector<double> h_fd, h_td, h_sync((size_t)n_cells * hf_cap * 4), h_ce;
  if (ac_fd) { h_fd.resize(C4 * rs_cap * 24); HIPCHK(c, hipMemcpyAsync(h_fd.data(), c->trk_acfd, sizeof(double) * h_fd.size(), hipMemcpyDeviceToHost, c->stream)); }
  if (ac_td) { h_td.resize(C4 * rs_cap * 144); HIPCHK(c, hipMemcpyAsync(h_td.data(), c->trk_actd, sizeof(double) * h_td.size(), hipMemcpyDeviceToHost, c->stream)); }
  if (sync_ce) { h_ce.resize((size_t)n_cells * hf_cap * 144); HIPCHK(c, hipMemcpyAsync(h_ce.data(), c->trk_syncce, sizeof(double) * h_ce.size(), hipMemcpyDeviceToHost, c->stream)); }
  HIPCHK(c, hipMemcpyAsync(h_sync.data(), c->trk_sync, sizeof(double) * h_sync.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(h_nmeas.data(), d_nmeas, sizeof(int) * C4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(h_cells.data(), c->trk_cells, sizeof(lcs_track_cell) * n_cells, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  rc = LCS_OK;
  for (size_t q = 0; q < C4; ++q) {
    const int rows = std::min(h_nmeas[q], max_rs);
    if (h_nmeas[q] > max_rs) rc = LCS_ERR_OVERFLOW;
    if (ac_fd) std::memcpy(ac_fd + q * max_rs * 24, &h_fd[q * rs_cap * 24], sizeof(double) * 24 * rows);
    if (ac_td) std::memcpy(ac_td + q * max_rs * 144, &h_td[q * rs_cap * 144], sizeof(double) * 144 * rows);
  }
  for (int i = 0; i < n_cells; ++i) {
    const int n_symb = (h_cells[i].cp_type == LCS_CP_NORMAL) ? 7 : 6;
    int hf = 0;
    while (((hf >> 1) * 20 + (hf & 1) * 10) * n_symb + n_symb - 1 < n_sym) ++hf;
    if (n_hf) n_hf[i] = std::min(hf, max_hf);
    if (hf > max_hf) rc = LCS_ERR_OVERFLOW;
    const int rows = std::min(hf, max_hf);
    if (sync) std::memcpy(sync + (size_t)i * max_hf * 4, &h_sync[(size_t)i * hf_cap * 4], sizeof(double) * 4 * rows);
    if (sync_ce) std::memcpy(sync_ce + (size_t)i * max_hf * 144, &h_ce[(size_t)i * hf_cap * 144], sizeof(double) * 144 * rows);
  }
  if (rc) c->err = "more rows than the output arrays hold";
  return rc;
}

// ------------------------------------------------------------------------------------------ continuous tracking
// The reference's tracker thread never stops: the three-symbol window of filter_ce (ref src/tracker_thread.cpp:176-201),
// the interpolation between consecutive filtered reference symbols (:383-477), the 72-deep history of do_ac_td (:343-371)
// and the four-frame fifo of do_mib_decode (:552-745) all reach across any boundary one could cut the symbol stream at.
// lcs_track_block cuts it: each call starts from nothing.  lcs_track_stream_block removes the cut for a caller that
// delivers a cell's symbols block after block: the context keeps the last three-to-four frames of every stream -- their
// frequency-domain rows (get_fd's output) on the device, their metadata and the bulk phase at their first symbol on the
// host -- and every call processes [carried frames ++ new symbols] as one block that starts at a frame boundary, with the
// same kernels, so each output row is computed exactly as a single call over the whole stream would compute it (nothing in
// the pipeline reaches back further than the carried frames), and hands out only the rows no earlier call could: every
// filtered reference symbol, channel-estimate row, autocorrelation row and MIB attempt exactly once, under its index in the
// whole stream.  The carried frames are not transformed again (k_trk_fd starts behind them; the rows are the ones the
// previous call computed, bit for bit) and frame offsets an earlier call attempted are not decoded again (k_trk_mib skips
// them); the per-port passes of k_trk_ce still run over them (the raw estimates and filter windows they rebuild are cheap).
namespace {
struct TrkStreamCell {
  std::vector<double> fo, ft, late;          // metadata of the carried symbols
  double bpo_before_tail = 0;                // bulk phase before the first carried symbol
  long long tail_start = 0;                  // stream index of the first carried symbol (a frame boundary)
  long long n_seen = 0;                      // symbols delivered so far
  long long next_raw[4] = {1, 1, 1, 1};      // per port: reference-symbol row (counted from the stream start) whose filter is emitted next
  long long ce_upto[4] = {0, 0, 0, 0};       // per port: channel estimates emitted for symbols below this
  long long mib_next = 0;                    // first frame offset not attempted yet
  int cp_type = 0, n_id_1 = -1, n_id_2 = -1, n_ports = 0;
};
// The carried symbols themselves stay on the DEVICE, already transformed: per CP type the frequency-domain rows
// [cells of that type][carried symbols][72] of the previous call (round 3 kept the time-domain samples on the host and ran
// them through get_fd again with every call).
struct TrkStream {
  std::vector<TrkStreamCell> cells;
  // indexed by LCS_CP_NORMAL / LCS_CP_EXTENDED: the carried rows, and a second buffer the next call's tail is written into
  // before the two change places at the commit (no allocation or hipFree -- a device-wide synchronisation -- per call)
  double2 *d_tail[3] = {nullptr, nullptr, nullptr}, *d_spare[3] = {nullptr, nullptr, nullptr};
  size_t tail_cap[3] = {0, 0, 0}, spare_cap[3] = {0, 0, 0};      // capacities in double2
};
}  // namespace

void lcs_track_stream_free(lcs_ctx *c) {
  TrkStream *st = static_cast<TrkStream *>(c->trk_stream);
  if (st) {
    for (double2 *p : st->d_tail) if (p) (void)hipFree(p);
    for (double2 *p : st->d_spare) if (p) (void)hipFree(p);
  }
  delete st;
  c->trk_stream = nullptr;
}

extern "C" int lcs_track_stream_reset(lcs_ctx *c) {
  if (!c) return LCS_ERR_BAD_ARG;
  lcs_track_stream_free(c);
  return LCS_OK;
}

extern "C" int lcs_track_stream_block(lcs_ctx *c, lcs_track_cell *cells, int n_cells, int n_sym, const void *td,
                                      const double *freq_off, const double *frame_timing, const double *late, double fc_requested,
                                      double fc_programmed, double fs_programmed, double *syms, double *ce, double *ce_pw, int ce_cap,
                                      int64_t *ce_from, int32_t *ce_n, double *meas, double *ac_fd, double *ac_td, int max_rs,
                                      int32_t *n_meas, int32_t *mib_ok, uint64_t *mib_bits, int max_off, int64_t *mib_from, int32_t *n_mib) {
  if (!c) return LCS_ERR_BAD_ARG;
  if (!cells || !td || !freq_off || !frame_timing || !late || n_cells < 1 || n_sym < 1 || max_rs < 0 || max_off < 0 || ce_cap < 0 ||
      ((ce || ce_pw) && (!ce_from || !ce_n)) || ((meas || ac_fd || ac_td) && !n_meas) || ((mib_ok || mib_bits) && (!mib_from || !n_mib))) {
    c->err = "bad argument";
    return LCS_ERR_BAD_ARG;
  }
  TrkStream *st = static_cast<TrkStream *>(c->trk_stream);
  if (!st) { st = new TrkStream(); c->trk_stream = st; }
  if (st->cells.empty()) {
    st->cells.resize(n_cells);
    for (int i = 0; i < n_cells; ++i) {
      TrkStreamCell &sc = st->cells[i];
      sc.cp_type = cells[i].cp_type; sc.n_id_1 = cells[i].n_id_1; sc.n_id_2 = cells[i].n_id_2; sc.n_ports = cells[i].n_ports;
      sc.bpo_before_tail = cells[i].bulk_phase_offset;
    }
  }
  if ((int)st->cells.size() != n_cells) { c->err = "the stream was started with a different number of cells: lcs_track_stream_reset first"; return LCS_ERR_BAD_ARG; }
  for (int i = 0; i < n_cells; ++i) {
    const TrkStreamCell &sc = st->cells[i];
    if (sc.cp_type != cells[i].cp_type || sc.n_id_1 != cells[i].n_id_1 || sc.n_id_2 != cells[i].n_id_2 || sc.n_ports != cells[i].n_ports) {
      c->err = "a tracked cell changed identity inside a stream: lcs_track_stream_reset first";
      return LCS_ERR_BAD_ARG;
    }
  }
  const double *tdv = static_cast<const double *>(td);
  int rc_all = LCS_OK;
  HIPCHK(c, hipSetDevice(c->device));
  // The stream's state is committed only after EVERY group of cells went through (round-3 advisory: a failure in the second
  // group used to leave the first one advanced): the new per-cell records and device tails are built aside.
  std::vector<TrkStreamCell> next = st->cells;
  bool group_done[3] = {false, false, false};
  auto fail = [&](int rc) { return rc; };              // nothing to undo: the new tails sit in the spare buffers until the commit
  // cells of one CP type carry the same number of frames: one extended block per CP type
  for (int cp = LCS_CP_NORMAL; cp <= LCS_CP_EXTENDED; ++cp) {
    std::vector<int> idx;
    for (int i = 0; i < n_cells; ++i) if (cells[i].cp_type == cp) idx.push_back(i);
    if (idx.empty()) continue;
    const int G = (int)idx.size(), F = (cp == LCS_CP_NORMAL) ? 140 : 120;
    const int n_tail = (int)(st->cells[idx[0]].n_seen - st->cells[idx[0]].tail_start);
    const int L = n_tail + n_sym;
    const long long T = st->cells[idx[0]].tail_start;
    for (int g = 0; g < G; ++g)
      if (st->cells[idx[g]].n_seen - st->cells[idx[g]].tail_start != n_tail || st->cells[idx[g]].tail_start != T) { c->err = "streams out of step"; return fail(LCS_ERR_BAD_ARG); }
    std::vector<lcs_track_cell> gc(G);
    std::vector<double> x_fo((size_t)G * L), x_ft((size_t)G * L), x_late((size_t)G * L);
    std::vector<int> mib_first(G);
    int rc = trk_ensure_ws(c, G, L);
    if (rc != LCS_OK) return fail(rc);
    // the carried rows go back to the head of every cell's rows, the new samples behind them
    if (n_tail > 0)
      HIPCHK(c, hipMemcpy2DAsync(c->trk_syms, sizeof(double2) * 72 * (size_t)L, st->d_tail[cp], sizeof(double2) * 72 * (size_t)n_tail,
                                 sizeof(double2) * 72 * (size_t)n_tail, G, hipMemcpyDeviceToDevice, c->stream));
    // the new samples: the group's cells are consecutive in td when the stream has one CP type (the usual case) -- then ONE strided
    // copy places them all behind their carried rows (64 cells: 64 copies of 2 MB each were 0.3-0.5 ms of a call's host and
    // copy-engine time); cells of a mixed stream go one by one
    bool one_copy = true;
    for (int g = 1; g < G; ++g) one_copy = one_copy && idx[g] == idx[0] + g;
    if (one_copy)
      HIPCHK(c, hipMemcpy2DAsync(c->trk_td + (size_t)n_tail * 128, sizeof(double2) * 128 * (size_t)L, tdv + (size_t)idx[0] * n_sym * 256,
                                 sizeof(double2) * 128 * (size_t)n_sym, sizeof(double2) * 128 * (size_t)n_sym, G, hipMemcpyDefault, c->stream));
    for (int g = 0; g < G; ++g) {
      const TrkStreamCell &sc = st->cells[idx[g]];
      gc[g] = cells[idx[g]];
      gc[g].bulk_phase_offset = sc.bpo_before_tail;
      // (hipMemcpyDefault: td may be pageable host memory, page-locked host memory -- DMA'd in place -- or device memory)
      if (!one_copy)
        HIPCHK(c, hipMemcpyAsync(c->trk_td + ((size_t)g * L + n_tail) * 128, tdv + (size_t)idx[g] * n_sym * 256, sizeof(double2) * 128 * (size_t)n_sym,
                                 hipMemcpyDefault, c->stream));
      auto join = [&](const std::vector<double> &tail, const double *fresh, std::vector<double> &out) {
        std::copy(tail.begin(), tail.end(), out.begin() + (size_t)g * L);
        std::copy(fresh + (size_t)idx[g] * n_sym, fresh + (size_t)(idx[g] + 1) * n_sym, out.begin() + (size_t)g * L + n_tail);
      };
      join(sc.fo, freq_off, x_fo); join(sc.ft, frame_timing, x_ft); join(sc.late, late, x_late);
      mib_first[g] = (int)std::max<long long>(0, sc.mib_next - T / F);
    }
    const int rs_cap = L / 3 + 4, n_off = std::max(0, L / 120 - 3);
    std::vector<double> o_syms(syms ? (size_t)G * L * 144 : 0), o_ce(ce ? (size_t)G * 4 * L * 144 : 0), o_pw((ce_pw || ce) ? (size_t)G * 4 * L * 4 : 0);
    std::vector<double> o_meas((size_t)G * 4 * rs_cap * LCS_TRK_MEAS);
    std::vector<int32_t> o_upto((size_t)G * 4), o_nmeas((size_t)G * 4), o_ok((size_t)G * std::max(1, n_off));
    std::vector<uint64_t> o_bits((size_t)G * std::max(1, n_off));
    const TrkCarried carried = {n_tail, mib_first.data()};
    rc = trk_block(c, gc.data(), G, L, c->trk_td, 2, x_fo.data(), x_ft.data(), x_late.data(), fc_requested, fc_programmed, fs_programmed,
                   syms ? o_syms.data() : nullptr, ce ? o_ce.data() : nullptr, (ce || ce_pw) ? o_pw.data() : nullptr, o_upto.data(),
                   o_meas.data(), rs_cap, o_nmeas.data(), o_ok.data(), o_bits.data(), std::max(1, n_off), nullptr, &carried);
    if (rc != LCS_OK) return fail(rc);
    std::vector<double> o_fd, o_tdc;
    if (ac_fd || ac_td) {
      if (ac_fd) o_fd.resize((size_t)G * 4 * rs_cap * 24);
      if (ac_td) o_tdc.resize((size_t)G * 4 * rs_cap * 144);
      rc = lcs_track_stats(c, G, L, ac_fd ? o_fd.data() : nullptr, ac_td ? o_tdc.data() : nullptr, rs_cap, nullptr, nullptr, 0, nullptr);
      if (rc != LCS_OK && rc != LCS_ERR_OVERFLOW) return fail(rc);      // no PSS/SSS rows were asked for: their overflow is of no concern
    }
    // the next carried tail: whole frames from T_next on -- its frequency-domain rows into a fresh device buffer, the bulk
    // phase before its first symbol (= the value used at the symbol before it) in ONE strided copy
    const long long n_after = st->cells[idx[0]].n_seen + n_sym;
    const long long T_next = std::max<long long>(0, (n_after - 3 * F) / F * F);
    const size_t keep_from = (size_t)(T_next - T), n_keep = (size_t)(n_after - T_next);
    std::vector<double> bpo_at(G, 0.0);
    {
      const size_t need = (size_t)72 * n_keep * G;
      if (need > st->spare_cap[cp]) {
        if (st->d_spare[cp]) { (void)hipFree(st->d_spare[cp]); st->d_spare[cp] = nullptr; st->spare_cap[cp] = 0; }
        const size_t cap = need + need / 4;               // head room: the tail's length varies by up to a frame from call to call
        const hipError_t e = hipMalloc((void **)&st->d_spare[cp], sizeof(double2) * cap);
        if (e != hipSuccess) { st->d_spare[cp] = nullptr; c->err = std::string("hipMalloc (carried symbols): ") + hipGetErrorString(e); return fail(LCS_ERR_HIP); }
        st->spare_cap[cp] = cap;
      }
      const TrkLayout Lay(c, G, L);
      hipError_t e2 = hipMemcpy2DAsync(st->d_spare[cp], sizeof(double2) * 72 * n_keep, c->trk_syms + keep_from * 72, sizeof(double2) * 72 * (size_t)L,
                                       sizeof(double2) * 72 * n_keep, G, hipMemcpyDeviceToDevice, c->stream);
      if (e2 == hipSuccess && T_next > T)
        e2 = hipMemcpy2DAsync(bpo_at.data(), sizeof(double), Lay.d_bpo + (size_t)(T_next - T - 1), sizeof(double) * (size_t)L, sizeof(double), G,
                              hipMemcpyDeviceToHost, c->stream);
      if (e2 == hipSuccess) e2 = hipStreamSynchronize(c->stream);
      if (e2 != hipSuccess) { c->err = std::string("carrying the stream's tail: ") + hipGetErrorString(e2); return fail(LCS_ERR_HIP); }
    }
    group_done[cp] = true;
    for (int g = 0; g < G; ++g) {
      const int i = idx[g];
      TrkStreamCell &sc = next[i];
      cells[i].bulk_phase_offset = gc[g].bulk_phase_offset;
      if (syms) std::memcpy(syms + (size_t)i * n_sym * 144, &o_syms[((size_t)g * L + n_tail) * 144], sizeof(double) * n_sym * 144);
      for (int p = 0; p < 4; ++p) {
        const size_t q = (size_t)g * 4 + p, Q = (size_t)i * 4 + p;
        // filtered reference symbols: row r of the block is raw row r + 1 of the block = raw row (rows before T) + r + 1 of the stream
        const long long rows_before = (T / F) * (p < 2 ? 40 : 20);
        int emitted = 0;
        for (int r = 0; r < o_nmeas[q]; ++r) {
          const long long raw = rows_before + r + 1;
          if (raw < sc.next_raw[p]) continue;
          if (emitted < max_rs) {
            if (meas) {
              double *m = meas + (Q * max_rs + emitted) * LCS_TRK_MEAS;
              std::memcpy(m, &o_meas[(q * rs_cap + r) * LCS_TRK_MEAS], sizeof(double) * LCS_TRK_MEAS);
              m[0] += (double)T;                               // symbol index in the stream
            }
            if (ac_fd) std::memcpy(ac_fd + (Q * max_rs + emitted) * 24, &o_fd[(q * rs_cap + r) * 24], sizeof(double) * 24);
            if (ac_td) std::memcpy(ac_td + (Q * max_rs + emitted) * 144, &o_tdc[(q * rs_cap + r) * 144], sizeof(double) * 144);
          } else rc_all = LCS_ERR_OVERFLOW;
          ++emitted;
          sc.next_raw[p] = raw + 1;
        }
        if (n_meas) n_meas[Q] = std::min(emitted, max_rs);
        // channel estimates: symbols [ce_upto, T + upto) are new
        const long long upto = (o_upto[q] > 0) ? T + o_upto[q] : sc.ce_upto[p];
        const long long from = sc.ce_upto[p];
        const int n_new = (int)std::max<long long>(0, upto - from);
        if (ce_from) ce_from[Q] = from;
        if (ce_n) ce_n[Q] = std::min(n_new, ce_cap);
        if (n_new > ce_cap && (ce || ce_pw)) rc_all = LCS_ERR_OVERFLOW;
        for (int r = 0; r < std::min(n_new, ce_cap); ++r) {
          const size_t src = q * L + (size_t)(from - T) + r;
          if (ce) std::memcpy(ce + (Q * ce_cap + r) * 144, &o_ce[src * 144], sizeof(double) * 144);
          if (ce_pw) std::memcpy(ce_pw + (Q * ce_cap + r) * 4, &o_pw[src * 4], sizeof(double) * 4);
        }
        if (upto > sc.ce_upto[p]) sc.ce_upto[p] = upto;
      }
      // MIB attempts: frame offset o of the block is offset T / F + o of the stream; an offset stays pending until it has been tried
      int em = 0;
      if (mib_from) mib_from[i] = sc.mib_next;
      for (int o = 0; o < n_off; ++o) {
        const long long og = T / F + o;
        if (og < sc.mib_next) continue;
        if (og > sc.mib_next || o_ok[(size_t)g * std::max(1, n_off) + o] == -1) break;
        if (em < max_off) {
          if (mib_ok) mib_ok[(size_t)i * max_off + em] = o_ok[(size_t)g * std::max(1, n_off) + o];
          if (mib_bits) mib_bits[(size_t)i * max_off + em] = o_bits[(size_t)g * std::max(1, n_off) + o];
        } else rc_all = LCS_ERR_OVERFLOW;
        ++em;
        sc.mib_next = og + 1;
      }
      if (n_mib) n_mib[i] = std::min(em, max_off);
      // the metadata of the carried symbols
      sc.n_seen = n_after;
      sc.fo.assign(x_fo.begin() + (size_t)g * L + keep_from, x_fo.begin() + (size_t)g * L + keep_from + n_keep);
      sc.ft.assign(x_ft.begin() + (size_t)g * L + keep_from, x_ft.begin() + (size_t)g * L + keep_from + n_keep);
      sc.late.assign(x_late.begin() + (size_t)g * L + keep_from, x_late.begin() + (size_t)g * L + keep_from + n_keep);
      if (T_next > T) sc.bpo_before_tail = bpo_at[g];
      sc.tail_start = T_next;
    }
  }
  // commit
  st->cells.swap(next);
  for (int cp = LCS_CP_NORMAL; cp <= LCS_CP_EXTENDED; ++cp)
    if (group_done[cp]) { std::swap(st->d_tail[cp], st->d_spare[cp]); std::swap(st->tail_cap[cp], st->spare_cap[cp]); }
  if (rc_all) c->err = "more rows than the output arrays hold (rows beyond the capacity were dropped)";
  return rc_all;
}
