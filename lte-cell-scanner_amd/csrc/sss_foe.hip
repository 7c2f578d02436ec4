// sss_foe.hip -- SSS maximum-likelihood detection and PSS/SSS fine frequency-offset estimate.
//
// Replaces extract_psss (ref src/searcher.cpp:516-530), sss_detect_getce_sss (:533-632),
// sss_detect_ml(_helper) (:636-693), sss_detect (:696-761) and pss_sss_foe (:767-850).
//
// All arithmetic in fp64 like the reference.  Per peak the work is tiny and latency-bound, so it
// is spread as widely as the data dependences allow:
//   k_sss_win    numbers the (buffer, peak) pairs of the whole batch (workgroup 0 writes the list for the kernels behind
//                it; k_peak_list does that alone in front of the pss_sss_foe stage entry point);
//                one workgroup per (peak, half-frame occurrence), one wave per 128-sample window
//                (PSS, extended-CP SSS, normal-CP SSS): frequency-correct while staging into LDS,
//                direct 62x128 DFT of the PSS/SSS subcarriers only (<= 60 windows per peak: an FFT
//                would buy nothing and the direct form is the more accurate one), channel
//                smoothing and noise power of the occurrence -> workspace record;
//   k_sss_ml     one workgroup per peak: even/odd combining in the reference's k order, the
//                168 x 2 x 2 ML search and the decision;
//   k_foe_win    one workgroup per (peak, occurrence): PSS and SSS windows, per-occurrence FOE term;
//   k_foe_fin    one thread per peak: sum the terms in occurrence order -> freq_fine.
// Every workgroup needs < 8 KB of LDS (k_sss_ml 17 KB) and at most 4 waves, so they can also be
// placed next to resident correlation workgroups of the following batch.
#include "lcs_internal.h"

#define SF_THREADS 256
#define MAX_HF 20
#define FS_LTE 30720000.0

#include "lte_device.h"

__device__ __forceinline__ int d_floor_i(double x) { return (int)floor(x); }
__device__ __forceinline__ double d_matlab_mod(double k, double n) { return (n == 0) ? k : (k - n * d_floor_i(k / n)); }
__device__ __forceinline__ double d_wrap(double x, double sm, double lg) { return d_matlab_mod(x - sm, lg - sm) + sm; }
__device__ __forceinline__ int d_range_len(double first, double incr, double last) {   // ref src/itpp_ext.cpp:97-109
  const double s1 = (double)((last - first > 0) - (last - first < 0));
  const double s2 = (double)((incr > 0) - (incr < 0));
  return (s1 * s2 >= 0) ? d_floor_i((last - first) / incr) + 1 : 0;
}

// workspace record of one (peak, occurrence), in doubles
#define SW_HSM 0       // 62 complex: smoothed PSS channel
#define SW_NRM 124     // 62 complex: normal-CP SSS bins
#define SW_EXT 248     // 62 complex: extended-CP SSS bins
#define SW_NP 372      // noise power of the occurrence
#define SW_ACC 374     // FOE: complex per-occurrence term
#define SW_REC 376
#define SW_ITEM ((size_t)MAX_HF * SW_REC)

// ---- 128-sample windows of the synchronisation signals, eight per wave (round 5; lte_device.h: fft128_x8) -----------------
// capbuf.mid(loc, 128) -> fshift(., foc_freq, fs) -> rotate left by 2 -> 128-point transform (ref :516-530): sample t of the
// window, rotated by cis(k t), is input n = (t - 2) & 127 of the transform.  Lane (w, l) holds inputs n = l + 8 j, i.e. samples
// t = (l + 2 + 8 j) & 127: cis(k t) = cis(k (l + 2)) cis(8 k j) -- one factor per lane and sixteen per window, which the
// window's eight lanes compute two each and share through LDS -- except where t wraps (j = 15, l >= 6: t = l - 6).  Four
// sincos per lane and 8 windows instead of sixteen (rounds 1-4: one per sample).
template <int KIND>
__device__ __forceinline__ void win_load16(const CapView &cap, long loc, int l, uint32_t n_cap, bool valid, cd2 (&x)[16]) {
  uint16_t r8[16];
  float2 r32[16];
  unsigned in_mask = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const long sidx = loc + ((l + 2 + 8 * j) & 127);
    const bool in = valid && sidx >= 0 && (uint64_t)sidx < n_cap;
    const size_t ci = in ? (size_t)sidx : 0;
    in_mask |= (in ? 1u : 0u) << j;
    if (KIND == 0) r8[j] = cap.c8[ci];
    else if (KIND == 1) r32[j] = cap.c32[ci];
    else { const double2 v = cap.c64[ci]; x[j] = mk(v.x, v.y); }
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (KIND == 0) { const uint32_t pr = r8[j]; x[j] = mk(-(double)(int)(int8_t)(pr & 255u) / 128.0, -(double)(int)(int8_t)(pr >> 8) / 128.0); }
    else if (KIND == 1) x[j] = mk((double)r32[j].x, (double)r32[j].y);
    if (!((in_mask >> j) & 1u)) x[j] = mk(0, 0);      // beyond the buffer: zeros (the reference's mid() would read out of bounds)
  }
}
// trot: this wave's [8][16] table in LDS.  The factors are computed BEFORE the samples are loaded (four sincos expansions with
// sixteen samples live beside them needed every register the wave can have).
// (cis_call is a real call, not inlined, lte_device.h: four inlined sincos expansions pushed k_sss_win to 256 + 26 registers -- 288 as allocated, more than the
// 284 a SIMD has free beside one resident correlation workgroup, tests/test_tables_abi.py)
struct WinRot { cd2 cu, cw; };
__device__ __forceinline__ WinRot win_rot_prepare(double k, int lane, cd2 *trot) {
  const int w = lane >> 3, l = lane & 7;
  WinRot r;
  r.cu = cis_call(k * (double)(l + 2));
  __builtin_amdgcn_sched_barrier(0);
  r.cw = cis_call(k * (double)(l - 6));                       // the wrapped sample of lanes l >= 6
  __builtin_amdgcn_sched_barrier(0);
  trot[w * 16 + l + 1] = cis_call(k * (double)(8 * (l + 1)));
  __builtin_amdgcn_sched_barrier(0);
  if (l < 7) trot[w * 16 + l + 9] = cis_call(k * (double)(8 * (l + 9)));
  lcs_wave_sync();
  return r;
}
__device__ __forceinline__ void win_rotate16(cd2 (&x)[16], const WinRot &r, int lane, const cd2 *trot) {
  const int w = lane >> 3, l = lane & 7;
  x[0] = cmul(x[0], r.cu);
#pragma unroll
  for (int j = 1; j < 15; ++j) x[j] = cmul(x[j], cmul(r.cu, trot[w * 16 + j]));
  x[15] = cmul(x[15], (l >= 6) ? r.cw : cmul(r.cu, trot[w * 16 + 15]));
}
// the 62 PSS / SSS bins [97..127, 1..31] among this lane's sixteen outputs X[(l + 8 c) + 16 k1] = x[8 c + k1]: index 0..61 or -1
__device__ __forceinline__ int win_bin62(int l, int q) {
  const int bin = (l + 8 * (q >> 3)) + 16 * (q & 7);
  return (bin >= 97) ? bin - 97 : ((bin >= 1 && bin <= 31) ? bin + 30 : -1);
}
// One window per lane group with the peak's rotation factors prepared before (win_rot_prepare: they depend on the peak's
// frequency only, so a wave computes them once per peak and keeps the table in LDS for all of the peak's occurrences)
template <int KIND>
__device__ __forceinline__ void win_fft8(const CapView &cap, long loc, bool valid, const WinRot &r, uint32_t n_cap, int lane, cd2 *tb, const cd2 *tw,
                                         const cd2 *trot, cd2 (&x)[16]) {
  const int l = lane & 7;
  win_load16<KIND>(cap, loc, l, n_cap, valid, x);
  win_rotate16(x, r, lane, trot);
  fft128_x8(x, tb, tw, lane);
}

// h_raw -> h_sm (13-tap mean, ref :584-588) for subcarrier t
__device__ __forceinline__ cd2 smooth13(const cd2 *h_raw, int t) {
  const int lt = (t - 6 > 0) ? t - 6 : 0, rt = (t + 6 < 61) ? t + 6 : 61;
  cd2 s = mk(0, 0);
  for (int i = lt; i <= rt; ++i) s = cadd(s, h_raw[i]);
  return cdivr(s, (double)(rt - lt + 1));
}
// sigpower(h_sm - h_raw) (ref :591), subcarrier order
__device__ __forceinline__ double noise_power(const cd2 *h_sm, const cd2 *h_raw) {
  double r = 0;
  for (int t = 0; t < 62; ++t) { const cd2 d = csub(h_sm[t], h_raw[t]); r += d.re * d.re + d.im * d.im; }
  return r / 62;
}

// ------------------------------------------------------------------ work list of peaks
// The peaks of a batch, numbered in (buffer, peak) order.  One wave: lane = capture buffer, 64 at a time.
__device__ __forceinline__ int peak_count(const int *__restrict__ npeaks, int n_buf, int s) {
  return (s < n_buf) ? min(max(npeaks[s], 0), LCS_MAXP) : 0;
}
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
  for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(v, off); if (lane >= off) v += u; }
  return v;
}
__device__ void peak_list_write(const int *__restrict__ npeaks, int n_buf, WorkItem *__restrict__ items, int *__restrict__ n_items, int lane) {
  int base = 0;
  for (int s0 = 0; s0 < n_buf; s0 += 64) {
    const int s = s0 + lane;
    const int cnt = peak_count(npeaks, n_buf, s);
    const int incl = wave_incl_scan(cnt, lane);
    const int at = base + incl - cnt;
    for (int p = 0; p < cnt; ++p) { items[at + p].slot = s; items[at + p].peak = p; }
    base += __shfl(incl, 63);
  }
  if (lane == 0) *n_items = base;
}
// the stage entry point of pss_sss_foe alone (no k_sss_win in front of it)
__global__ __launch_bounds__(64) void k_peak_list(const int *__restrict__ npeaks, int n_buf, WorkItem *__restrict__ items,
                                                  int *__restrict__ n_items) {
  LCS_TAIL_PRIO();
  peak_list_write(npeaks, n_buf, items, n_items, threadIdx.x);
}
// The same numbering without a list, for the kernel that runs right behind the peak search (round 2: a one-wave
// kernel in between): every wave calls these with all 64 lanes; results are wave-uniform.
__device__ __forceinline__ int peak_total(const int *__restrict__ npeaks, int n_buf, int lane) {
  int base = 0;
  for (int s0 = 0; s0 < n_buf; s0 += 64) base += __shfl(wave_incl_scan(peak_count(npeaks, n_buf, s0 + lane), lane), 63);
  return base;
}
__device__ __forceinline__ WorkItem peak_lookup(const int *__restrict__ npeaks, int n_buf, int it, int lane) {
  WorkItem w;
  w.slot = 0; w.peak = 0;
  int base = 0;
  for (int s0 = 0; s0 < n_buf; s0 += 64) {
    const int cnt = peak_count(npeaks, n_buf, s0 + lane);
    const int incl = wave_incl_scan(cnt, lane);
    const int total = __shfl(incl, 63);
    if (it < base + total) {                                            // wave-uniform
      const unsigned long long m = __ballot(it < base + incl);          // first lane whose range reaches past `it`
      const int src = __ffsll((long long)m) - 1;
      w.slot = s0 + src;
      w.peak = it - base - __shfl(incl - cnt, src);
      return w;
    }
    base += total;
  }
  return w;
}

// ------------------------------------------------------------------ sss_detect geometry
struct SssGeo { double peak_loc, k_factor, kph; int n_pss; };
__device__ __forceinline__ SssGeo sss_geometry(const lcs_cell &cell, const SlotParams &p, uint32_t n_cap) {
  SssGeo g;
  g.peak_loc = cell.ind;
  g.k_factor = (p.fc_req - cell.freq) / p.fc_prog;
  if (g.peak_loc + 9 < 162) g.peak_loc += 9600 * g.k_factor;
  g.n_pss = d_range_len(g.peak_loc, g.k_factor * 9600, (double)n_cap - 125 - 9);
  if (g.n_pss > MAX_HF) g.n_pss = MAX_HF;
  const double fs = p.fs_prog * g.k_factor;
  g.kph = M_PI * (-cell.freq) / (fs / 2);
  return g;
}

// One wave per workgroup; a job = TWO occurrences (k = 2 m, 2 m + 1) of one peak = 6 windows (window slot w: occurrence w / 3;
// kind w % 3 = PSS window, extended-CP SSS window, normal-CP SSS window, ref :578-597); slots 6, 7 idle.
#define SW_WAVES 4           // independent waves per workgroup (a workgroup then fills the slot of the correlation workgroup it displaces)
#define SW_THREADS (64 * SW_WAVES)
template <int KIND>      // the source format (one instantiation each: the register allocation of a kernel holding all three load paths is the widest one's)
__global__ __launch_bounds__(SW_THREADS) void k_sss_win(const lcs_cell *__restrict__ peaks, const int *__restrict__ npeaks, int n_buf,
                                                        WorkItem *__restrict__ items, int *__restrict__ n_items,
                                                        const CapSrc src,
                                                        uint32_t n_cap, const SlotParams *__restrict__ params,
                                                        const double2 *__restrict__ pss_fd, double *__restrict__ ws) {
  LCS_TAIL_PRIO();
  __shared__ cd2 tw[128];
  __shared__ cd2 tb_all[SW_WAVES][8 * FFT128_WSTRIDE];
  __shared__ cd2 trot_all[SW_WAVES][8 * 16];
  const int lane = threadIdx.x & 63, w = lane >> 3, l = lane & 7;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // wave-uniform, and the compiler is told so: the job's records load into SGPRs
  // a wave's transpose buffer also holds the PSS channel estimates after the transform
  cd2 *tb = tb_all[wv], *trot = trot_all[wv];
  cd2 (*h_raw)[62] = reinterpret_cast<cd2 (*)[62]>(tb), (*h_sm)[62] = reinterpret_cast<cd2 (*)[62]>(tb + 2 * 62);
  static_assert(4 * 62 <= 8 * FFT128_WSTRIDE, "the estimates fit the transpose buffer");
  fft128_twiddle_table(tw, threadIdx.x, SW_THREADS);
  __syncthreads();
  // the work list: numbered here from the per-buffer counts; workgroup 0 also writes it out for the kernels that follow
  if (blockIdx.x == 0 && wv == 0) peak_list_write(npeaks, n_buf, items, n_items, lane);
  const int n_pk = peak_total(npeaks, n_buf, lane);
  // a job = one PEAK: its record, geometry and rotation factors once, then its occurrences two at a time (k = k0, k0 + 1: window
  // slot w -> occurrence w / 3; kind w % 3 = PSS window, extended-CP SSS window, normal-CP SSS window, ref :578-597; slots 6, 7 idle)
  // (a handful of peaks -- one buffer, the streaming mode -- are better served by latency: then a job is ONE pair of occurrences)
  const int split = (n_pk <= 64) ? MAX_HF / 2 : 1;
  for (int job = blockIdx.x * SW_WAVES + wv; job < n_pk * split; job += gridDim.x * SW_WAVES) {
    const int it = job / split, part = job - it * split;
    const WorkItem wi = peak_lookup(npeaks, n_buf, it, lane);
    const int slot = __builtin_amdgcn_readfirstlane(wi.slot), pk = __builtin_amdgcn_readfirstlane(wi.peak);
    const lcs_cell cell = peaks[(size_t)slot * LCS_MAXP + pk];
    const SlotParams p = params[slot];
    const SssGeo g = sss_geometry(cell, p, n_cap);
    const CapView cap = cap_view(src, slot);
    const WinRot rot = win_rot_prepare(g.kph, lane, trot);
    const int occ = w / 3, kind = w - 3 * occ;
    const int k_first = (split == 1) ? 0 : 2 * part, k_last = (split == 1) ? g.n_pss : min(g.n_pss, 2 * part + 2);
    for (int k0 = k_first; k0 < k_last; k0 += 2) {
      const int k = k0 + occ;
      const bool valid = w < 6 && k < g.n_pss;
      const uint32_t pss_loc = (uint32_t)d_round_i(g.peak_loc + k * (g.k_factor * 9600));
      const long pss_dft = (long)(pss_loc + 9 - 2);
      const long loc = (kind == 0) ? pss_dft : (kind == 1 ? pss_dft - 128 - 32 : pss_dft - 128 - 9);
      cd2 x[16];
      win_fft8<KIND>(cap, loc, valid, rot, n_cap, lane, tb, tw, trot, x);
      double *rec = ws + (size_t)it * SW_ITEM + (size_t)k * SW_REC;
      if (valid) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int b = win_bin62(l, q);
          if (b < 0) continue;
          const cd2 o = cdivr(x[q], sqrt(128.0));                         // /sqrt(128) (ref :527-529)
          if (kind == 0) { const double2 f = pss_fd[cell.n_id_2 * 62 + b]; h_raw[occ][b] = cmul(o, mk(f.x, -f.y)); }
          else { rec[(kind == 1 ? SW_EXT : SW_NRM) + 2 * b] = o.re; rec[(kind == 1 ? SW_EXT : SW_NRM) + 2 * b + 1] = o.im; }
        }
      }
      lcs_wave_sync();
      for (int o2 = 0; o2 < 2; ++o2) {
        if (k0 + o2 >= g.n_pss) break;
        double *r2 = ws + (size_t)it * SW_ITEM + (size_t)(k0 + o2) * SW_REC;
        if (lane < 62) {
          const cd2 v = smooth13(h_raw[o2], lane);
          h_sm[o2][lane] = v;
          r2[SW_HSM + 2 * lane] = v.re; r2[SW_HSM + 2 * lane + 1] = v.im;
        }
      }
      lcs_wave_sync();
      if (lane < 2 && k0 + lane < g.n_pss) ws[(size_t)it * SW_ITEM + (size_t)(k0 + lane) * SW_REC + SW_NP] = noise_power(h_sm[lane], h_raw[lane]);
      lcs_wave_sync();                                   // the buffer is rewritten by the next pair of occurrences
    }
    lcs_wave_sync();                                     // the rotation table is rewritten by the wave's next peak
  }
}

// ------------------------------------------------------------------ combining, ML, decision
struct MlShared {
  double np12[124];
  double rnp12[124];          // 1/np12
  cd2 nrm12[124];
  cd2 ext12[124];
  double ll[2][2][168];       // [nrm/ext][column][n_id_1]
  double dec[4][4];
};

__global__ __launch_bounds__(SF_THREADS) void k_sss_ml(lcs_cell *__restrict__ peaks, const WorkItem *__restrict__ items,
                                                       const int *__restrict__ n_items, uint32_t n_cap,
                                                       const SlotParams *__restrict__ params, double thresh2,
                                                       const int8_t *__restrict__ sss_fd, const double *__restrict__ ws,
                                                       double *dbg) {
  LCS_TAIL_PRIO();
  __shared__ MlShared S;
  const int tid = threadIdx.x;
  for (int it = blockIdx.x; it < *n_items; it += gridDim.x) {
    const int slot = items[it].slot;
    lcs_cell *cell_p = peaks + (size_t)slot * LCS_MAXP + items[it].peak;
    const lcs_cell cell = *cell_p;
    const SlotParams p = params[slot];
    const SssGeo g = sss_geometry(cell, p, n_cap);
    if (g.n_pss < 1) continue;
    const int n_id_2 = cell.n_id_2;
    const double *wsi = ws + (size_t)it * SW_ITEM;
    __syncthreads();
    // combine even (h1) / odd (h2) occurrences per subcarrier (ref :618-631)
    if (tid < 124) {
      const int h = tid / 62, t = tid % 62;
      double s = 0;
      cd2 sn = mk(0, 0), se = mk(0, 0);
      for (int k = h; k < g.n_pss; k += 2) {
        const double *rec = wsi + (size_t)k * SW_REC;
        const cd2 hs = mk(rec[SW_HSM + 2 * t], rec[SW_HSM + 2 * t + 1]);
        const double rnp = 1.0 / rec[SW_NP];
        s += cabs2(hs) * rnp;
        const cd2 w = cmul(cconj(hs), mk(rnp, 0));
        sn = cadd(sn, cmul(w, mk(rec[SW_NRM + 2 * t], rec[SW_NRM + 2 * t + 1])));
        se = cadd(se, cmul(w, mk(rec[SW_EXT + 2 * t], rec[SW_EXT + 2 * t + 1])));
      }
      const double np_est = 1 / (1 + s);
      S.np12[tid] = np_est;
      S.rnp12[tid] = 1.0 / np_est;
      S.nrm12[tid] = cscale(sn, np_est);
      S.ext12[tid] = cscale(se, np_est);
    }
    __syncthreads();
    // ML over 168 n_id_1 x {12,21} x {nrm,ext} (ref :636-693)
    for (int job = tid; job < 168 * 4; job += SF_THREADS) {
      const int n1 = job >> 2, col = job & 1, ext = (job >> 1) & 1;
      const cd2 *est = ext ? S.ext12 : S.nrm12;
      const int8_t *h1 = sss_fd + ((n1 * 3 + n_id_2) * 2 + 0) * 62;
      const int8_t *h2 = sss_fd + ((n1 * 3 + n_id_2) * 2 + 1) * 62;
      const int8_t *first = col ? h2 : h1, *second = col ? h1 : h2;
      cd2 acc = mk(0, 0);
      for (int i = 0; i < 124; ++i) {
        const double tv = (double)(i < 62 ? first[i] : second[i - 62]);
        acc = cadd(acc, cmul(cconj(est[i]), mk(tv, 0)));
      }
      const double ang = atan2_call(acc.im, acc.re);
      const cd2 rot = cis_call(-ang);
      double s1 = 0, s2 = 0;
      for (int i = 0; i < 124; ++i) {      // the two sums of ref :649 keep their own order; x/np as x*(1/np)
        const double tv = (double)(i < 62 ? first[i] : second[i - 62]);
        const cd2 d = csub(cmul(mk(tv, 0), rot), est[i]);
        s1 += (d.re * d.re) * S.rnp12[i];
        s2 += (d.im * d.im) * S.rnp12[i];
      }
      S.ll[ext][col][n1] = -s1 - s2;
    }
    __syncthreads();
    // decision (ref :719-758): lanes 0..3 each scan one of the four likelihood columns (max, first
    // arg-max, sum, sum of squares in index order); lane 0 combines them in the reference's order
    if (tid < 4) {
      const double *col = &S.ll[tid >> 1][tid & 1][0];
      double mx = col[0], sum = 0, sq = 0;
      int am = 0;
      for (int t = 0; t < 168; ++t) {
        const double v = col[t];
        if (v > mx) { mx = v; am = t; }
        sum += v; sq += v * v;
      }
      S.dec[tid][0] = mx; S.dec[tid][1] = (double)am; S.dec[tid][2] = sum; S.dec[tid][3] = sq;
    }
    __syncthreads();
    if (tid == 0) {
      const double mx_n = (S.dec[1][0] > S.dec[0][0]) ? S.dec[1][0] : S.dec[0][0];
      const double mx_e = (S.dec[3][0] > S.dec[2][0]) ? S.dec[3][0] : S.dec[2][0];
      const int e = (mx_n > mx_e) ? 0 : 1;
      const int cp_type = e ? LCS_CP_EXTENDED : LCS_CP_NORMAL;
      const double mx0 = S.dec[2 * e][0], mx1 = S.dec[2 * e + 1][0];
      const double k_factor = g.k_factor;
      double frame_start = cell.ind + (128 + 9 - 960 - 2) * 16 / FS_LTE * p.fs_prog * k_factor;
      int col;
      if (mx0 > mx1) col = 0;
      else { col = 1; frame_start = frame_start + 9600 * k_factor * 16 / FS_LTE * p.fs_prog * k_factor; }   // k_factor^2: quirk Q3
      frame_start = d_wrap(frame_start, -0.5, (2 * 9600.0 - 0.5) * 16 / FS_LTE * p.fs_prog * k_factor);
      const int n_id_1_est = (int)S.dec[2 * e + col][1];
      const double lik_final = S.dec[2 * e + col][0];
      double sum = 0, sq = 0;
      for (int q = 0; q < 4; ++q) { sum += S.dec[q][2]; sq += S.dec[q][3]; }
      const int len = 672;
      const double lik_mean = sum / len;
      const double lik_var = (sq - sum * sum / len) / (len - 1);    // itpp::variance (unbiased)
      if (lik_final >= lik_mean + sqrt(lik_var) * thresh2) {
        cell_p->n_id_1 = n_id_1_est;
        cell_p->cp_type = cp_type;
        cell_p->frame_start = frame_start;
      }
    }
    if (dbg) {   // the reference's "only used for testing" outputs
      for (int i = tid; i < 62; i += SF_THREADS) {
        dbg[i] = S.np12[i]; dbg[62 + i] = S.np12[62 + i];
        dbg[124 + 2 * i] = S.nrm12[i].re; dbg[124 + 2 * i + 1] = S.nrm12[i].im;
        dbg[248 + 2 * i] = S.nrm12[62 + i].re; dbg[248 + 2 * i + 1] = S.nrm12[62 + i].im;
        dbg[372 + 2 * i] = S.ext12[i].re; dbg[372 + 2 * i + 1] = S.ext12[i].im;
        dbg[496 + 2 * i] = S.ext12[62 + i].re; dbg[496 + 2 * i + 1] = S.ext12[62 + i].im;
      }
      for (int i = tid; i < 168 * 2; i += SF_THREADS) {
        dbg[620 + i] = S.ll[0][i & 1][i >> 1];           // log_lik_nrm [168][2]
        dbg[620 + 336 + i] = S.ll[1][i & 1][i >> 1];     // log_lik_ext [168][2]
      }
    }
  }
}

// ------------------------------------------------------------------ pss_sss_foe
struct FoeGeo { int pss_sss_dist, sn_init, n_sss; double first_sss, step, k_factor, kph; bool ok; };
__device__ __forceinline__ FoeGeo foe_geometry(const lcs_cell &cell, const SlotParams &p, uint32_t n_cap) {
  FoeGeo g;
  g.ok = false;
  g.n_sss = 0;
  if (cell.n_id_1 < 0) return g;
  const double k_factor = (p.fc_req - cell.freq) / p.fc_prog;
  g.k_factor = k_factor;
  if (cell.cp_type == LCS_CP_NORMAL) {
    g.pss_sss_dist = (int)(uint16_t)d_round_i((128 + 9) * 16 / FS_LTE * p.fs_prog * k_factor);
    g.first_sss = cell.frame_start + (960 - 128 - 9 - 128) * 16 / FS_LTE * p.fs_prog * k_factor;
  } else if (cell.cp_type == LCS_CP_EXTENDED) {
    g.pss_sss_dist = (int)(uint16_t)d_round_i((128 + 32) * k_factor);   // quirk Q4
    g.first_sss = cell.frame_start + (960 - 128 - 32 - 128) * 16 / FS_LTE * p.fs_prog * k_factor;
  } else return g;
  g.first_sss = d_wrap(g.first_sss, -0.5, 9600 * 2 - 0.5);
  if (g.first_sss - 9600 * k_factor > -0.5) { g.first_sss -= 9600 * k_factor; g.sn_init = 10; } else g.sn_init = 0;
  g.step = 9600 * 16 / FS_LTE * p.fs_prog * k_factor;
  g.n_sss = d_range_len(g.first_sss, g.step, (double)((int)n_cap - 127 - g.pss_sss_dist - 100));
  if (g.n_sss > MAX_HF) g.n_sss = MAX_HF;
  const double fs = p.fs_prog * k_factor;
  g.kph = M_PI * (-cell.freq) / (fs / 2);
  g.ok = true;
  return g;
}

// One wave per workgroup; a job = FOUR occurrences of one peak = 8 windows (window slot w: occurrence w >> 1; w & 1 = 0: the
// PSS window, 1: the SSS window in front of it, ref :803-845).
#define FW_WAVES 4
#define FW_THREADS (64 * FW_WAVES)
#define FW_QUADS (MAX_HF / 4)
template <int KIND>
__global__ __launch_bounds__(FW_THREADS) void k_foe_win(const lcs_cell *__restrict__ peaks, const WorkItem *__restrict__ items,
                                                        const int *__restrict__ n_items,
                                                        const CapSrc src,
                                                        uint32_t n_cap, const SlotParams *__restrict__ params,
                                                        const double2 *__restrict__ pss_fd, const int8_t *__restrict__ sss_fd,
                                                        double *__restrict__ ws) {
  LCS_TAIL_PRIO();
  __shared__ cd2 tw[128];
  __shared__ cd2 tb_all[FW_WAVES][8 * FFT128_WSTRIDE];
  __shared__ cd2 trot_all[FW_WAVES][8 * 16];
  __shared__ cd2 h_sm_all[FW_WAVES][4][62];
  const int lane = threadIdx.x & 63, w = lane >> 3, l = lane & 7;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // wave-uniform, and the compiler is told so: the job's records load into SGPRs
  cd2 *tb = tb_all[wv], *trot = trot_all[wv];            // (the transpose buffer also holds the estimates after the transform)
  cd2 (*h_raw)[62] = reinterpret_cast<cd2 (*)[62]>(tb), (*aux)[62] = reinterpret_cast<cd2 (*)[62]>(tb + 4 * 62), (*h_sm)[62] = h_sm_all[wv];
  static_assert(8 * 62 <= 8 * FFT128_WSTRIDE, "the estimates fit the transpose buffer");
  fft128_twiddle_table(tw, threadIdx.x, FW_THREADS);
  __syncthreads();
  // a job = FOUR occurrences of one peak (cell) = 8 windows (window slot w -> occurrence w >> 1; w & 1 = 0: the PSS window, 1: the
  // SSS window in front of it, ref :803-845).  (One wave per cell walking all its occurrences -- the form k_sss_win takes -- was
  // measured slower here: a cell has only four such jobs, and a batch only a few hundred cells.)
  const int n_jobs = *n_items * FW_QUADS;
  for (int job = blockIdx.x * FW_WAVES + wv; job < n_jobs; job += gridDim.x * FW_WAVES) {
    const int it = job / FW_QUADS, k0 = 4 * (job % FW_QUADS);
    const int slot = items[it].slot;
    const lcs_cell cell = peaks[(size_t)slot * LCS_MAXP + items[it].peak];
    const SlotParams p = params[slot];
    const FoeGeo g = foe_geometry(cell, p, n_cap);
    if (!g.ok || k0 >= g.n_sss) continue;
    const CapView cap = cap_view(src, slot);
    const WinRot rot = win_rot_prepare(g.kph, lane, trot);
    // exp(J*pi*-freq/(FS_LTE/16/2)*-pss_sss_dist), evaluated left to right (ref :832)
    double ph_im = M_PI;
    ph_im = ph_im * (-cell.freq);
    ph_im = ph_im / (FS_LTE / 16 / 2);
    ph_im = ph_im * (double)(-g.pss_sss_dist);
    const cd2 ph = cis(ph_im);
    const int occ = w >> 1, is_sss = w & 1;
    {
      const int k = k0 + occ;
      const bool valid = k < g.n_sss;
      const uint32_t sss_loc = (uint32_t)d_round_i(g.first_sss + k * g.step);
      const long loc = is_sss ? (long)sss_loc : (long)(sss_loc + g.pss_sss_dist);
      cd2 x[16];
      win_fft8<KIND>(cap, loc, valid, rot, n_cap, lane, tb, tw, trot, x);
      if (valid) {
        // the slot number toggles with every occurrence, starting from sn_init (ref :800, :813)
        const int sn = ((k & 1) == 0) ? g.sn_init : 10 - g.sn_init;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int b = win_bin62(l, q);
          if (b < 0) continue;
          const cd2 o = cdivr(x[q], sqrt(128.0));
          if (!is_sss) { const double2 f = pss_fd[cell.n_id_2 * 62 + b]; h_raw[occ][b] = cmul(o, mk(f.x, -f.y)); }
          else {
            const double sf = (double)sss_fd[((cell.n_id_1 * 3 + cell.n_id_2) * 2 + (sn != 0)) * 62 + b];
            aux[occ][b] = cmul(cmul(o, ph), mk(sf, 0));
          }
        }
      }
      lcs_wave_sync();
      for (int o2 = 0; o2 < 4; ++o2)
        if (k0 + o2 < g.n_sss && lane < 62) h_sm[o2][lane] = smooth13(h_raw[o2], lane);
      lcs_wave_sync();
      if (lane < 4 && k0 + lane < g.n_sss) {     // sum over the 62 subcarriers, in subcarrier order (ref :836-843)
        const double np = noise_power(h_sm[lane], h_raw[lane]);
        cd2 acc = mk(0, 0);
        for (int t = 0; t < 62; ++t) {
          const double a2 = cabs2(h_sm[lane][t]);
          const double wgt = a2 * (1.0 / (2 * a2 * np + np * np));
          acc = cadd(acc, cmul(cmul(cconj(aux[lane][t]), h_raw[lane][t]), mk(wgt, 0)));
        }
        double *rec = ws + (size_t)it * SW_ITEM + (size_t)(k0 + lane) * SW_REC;
        rec[SW_ACC] = acc.re; rec[SW_ACC + 1] = acc.im;
      }
      lcs_wave_sync();                                   // the buffers are rewritten by the wave's next job
    }
  }
}

__global__ __launch_bounds__(64) void k_foe_fin(lcs_cell *__restrict__ peaks, const WorkItem *__restrict__ items,
                                                const int *__restrict__ n_items, uint32_t n_cap,
                                                const SlotParams *__restrict__ params, const double *__restrict__ ws) {
  LCS_TAIL_PRIO();
  const int it = blockIdx.x * blockDim.x + threadIdx.x;
  if (it >= *n_items) return;
  const int slot = items[it].slot;
  lcs_cell *cell_p = peaks + (size_t)slot * LCS_MAXP + items[it].peak;
  const lcs_cell cell = *cell_p;
  const SlotParams p = params[slot];
  const FoeGeo g = foe_geometry(cell, p, n_cap);
  if (!g.ok) return;
  cd2 M = mk(0, 0);
  for (int k = 0; k < g.n_sss; ++k) {
    const double *rec = ws + (size_t)it * SW_ITEM + (size_t)k * SW_REC;
    M = cadd(M, mk(rec[SW_ACC], rec[SW_ACC + 1]));
  }
  cell_p->freq_fine = cell.freq + atan2(M.im, M.re) / (2 * M_PI) / (1 / (p.fs_prog * g.k_factor) * g.pss_sss_dist);
}

// ------------------------------------------------------------------ launchers
// mode bit 0: run sss_detect, bit 1: run pss_sss_foe (only for cells whose SSS was found)
static int run_sss_foe(lcs_ctx *c, int n_buf, uint32_t n_cap, double thresh2, int mode, double *dbg) {
  const size_t cap_items = (size_t)n_buf * LCS_MAXP;
  if (cap_items > c->sss_ws_items) {
    if (c->sss_ws) { HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->sss_ws); c->sss_ws = nullptr; }
    if (c->pk_items) { (void)hipFree(c->pk_items); c->pk_items = nullptr; }
    HIPCHK(c, hipMalloc((void **)&c->sss_ws, cap_items * SW_ITEM * sizeof(double)));
    HIPCHK(c, hipMalloc((void **)&c->pk_items, cap_items * sizeof(WorkItem)));
    if (!c->n_pk) HIPCHK(c, hipMalloc((void **)&c->n_pk, 4 * sizeof(int)));
    c->sss_ws_items = cap_items;
  }
  const CapSrc src = lcs_cap_src(c, n_cap);
  // enough workgroups for every (peak, occurrence) of a typical batch to be resident at once; the
  // kernels loop over the work list, so larger batches only take more rounds
  const int win_grid = (int)std::min<size_t>((cap_items + 3) / 4, LCS_WIN_GRID);      // a wave per peak, four waves per workgroup
  const int item_grid = (int)std::min<size_t>(cap_items, LCS_ITEM_GRID);
  if (!(mode & 1)) hipLaunchKernelGGL(k_peak_list, dim3(1), dim3(64), 0, c->stream, c->npeaks, n_buf, c->pk_items, c->n_pk);
  if (mode & 1) {
#define SSW_LAUNCH(KIND) hipLaunchKernelGGL(k_sss_win<KIND>, dim3(win_grid), dim3(SW_THREADS), 0, c->stream, c->peaks, c->npeaks, n_buf, c->pk_items, c->n_pk, src, \
                                            n_cap, c->params, c->d_pss_fd, c->sss_ws)
    if (src.c8) SSW_LAUNCH(0);
    else if (src.c32) SSW_LAUNCH(1);
    else SSW_LAUNCH(2);
#undef SSW_LAUNCH
    hipLaunchKernelGGL(k_sss_ml, dim3(item_grid), dim3(SF_THREADS), 0, c->stream, c->peaks, c->pk_items, c->n_pk, n_cap,
                       c->params, thresh2, c->d_sss_fd, c->sss_ws, dbg);
  }
  if (mode & 2) {
#define FOW_LAUNCH(KIND) hipLaunchKernelGGL(k_foe_win<KIND>, dim3((int)std::min<size_t>((cap_items * FW_QUADS + 3) / 4, LCS_WIN_GRID)), dim3(FW_THREADS), 0, c->stream, \
                                            c->peaks, c->pk_items, c->n_pk, src, n_cap, c->params, c->d_pss_fd, c->d_sss_fd, c->sss_ws)
    if (src.c8) FOW_LAUNCH(0);
    else if (src.c32) FOW_LAUNCH(1);
    else FOW_LAUNCH(2);
#undef FOW_LAUNCH
    hipLaunchKernelGGL(k_foe_fin, dim3((unsigned)((cap_items + 63) / 64)), dim3(64), 0, c->stream, c->peaks, c->pk_items,
                       c->n_pk, n_cap, c->params, c->sss_ws);
  }
  HIPCHK(c, hipGetLastError());
  return LCS_OK;
}

int lcs_launch_sss_foe(lcs_ctx *c, int n_buf, uint32_t n_cap, double thresh2_n_sigma, double *dbg) {
  return run_sss_foe(c, n_buf, n_cap, thresh2_n_sigma, 3, dbg);
}
// Single-cell helpers for the stage entry points: peaks[0] of slot 0 holds the cell (npeaks[0] = 1).
int lcs_launch_sss_only(lcs_ctx *c, uint32_t n_cap, double thresh2_n_sigma, double *dbg) {
  return run_sss_foe(c, 1, n_cap, thresh2_n_sigma, 1, dbg);
}
int lcs_launch_foe_only(lcs_ctx *c, uint32_t n_cap) { return run_sss_foe(c, 1, n_cap, 0.0, 2, nullptr); }
