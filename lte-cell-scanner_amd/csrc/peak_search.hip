// peak_search.hip -- greedy PSS peak search, one workgroup per capture buffer.
//
// Replaces peak_search (ref src/searcher.cpp:422-510).  The reference loop is inherently
// sequential (find global max -> threshold -> refine -> cancel -> repeat, a few dozen
// iterations at most); each iteration is a data-parallel pass over the 3x9600 working copy,
// so it runs as ONE 1024-thread workgroup per buffer and the chain never leaves the device.
// Semantics kept: per-row first arg-max then first row with the maximum (= smallest linear
// index among equal maxima), the uint16 refine loop that yields ind = -1 when
// peak_ind < ds_comb_arm (quirk Q2), +-274 cancellation in the peak's own row, the no-op
// "other PSS" loop (quirk Q1, omitted because it has no effect), the -12 dB floor.
#include "lcs_internal.h"

#define PS_THREADS 1024
#define PS_MAX_ITER 4096

__device__ __forceinline__ void cell_init(lcs_cell &c) {
  c.fc_requested = __longlong_as_double(0x7ff8000000000000LL);
  c.fc_programmed = c.fc_requested; c.pss_pow = c.fc_requested; c.freq = c.fc_requested;
  c.frame_start = c.fc_requested; c.freq_fine = c.fc_requested; c.freq_superfine = c.fc_requested;
  c.ind = -1; c.n_id_2 = -1; c.n_id_1 = -1; c.cp_type = LCS_CP_UNKNOWN; c.n_ports = -1; c.n_rb_dl = -1;
  c.phich_duration = 0; c.phich_resource = 0; c.sfn = -1; c.reserved = 0;
}

// Stage entry point (lcs_peak_search: arbitrary doubles from the caller), working copy in global
// memory.  The fused chain uses k_peak_search_reg below.
typedef double WT;
__global__ __launch_bounds__(PS_THREADS) void k_peak_search(const double *__restrict__ pow_, const int *__restrict__ frq,
                                                             const double *__restrict__ zth,
                                                             const float *__restrict__ single,
                                                             const double *__restrict__ fset,
                                                             const SlotParams *__restrict__ params, double *work,
                                                             lcs_cell *__restrict__ peaks, int *__restrict__ npeaks,
                                                             XcGeom geo, double udb10_m12) {
  const int slot = blockIdx.x;
  const int tid = threadIdx.x;
  const int NE = 3 * LCS_N_IDX;
  const double *pw = pow_ + (size_t)slot * NE;
  const int *fq = frq + (size_t)slot * NE;
  const double *z = zth + (size_t)slot * LCS_N_IDX;
  const float *sg = single + (size_t)slot * geo.G * LCS_N_IDX * LCS_TG;   // group-major (pss_xcorr.hip)
  WT *wk = reinterpret_cast<WT *>(work + (size_t)slot * NE);
  lcs_cell *out = peaks + (size_t)slot * LCS_MAXP;

  __shared__ double s_val[PS_THREADS / 64];
  __shared__ int s_idx[PS_THREADS / 64];
  __shared__ double s_peak_pow;
  __shared__ int s_peak_lin;
  __shared__ int s_stop;

  for (int e = tid; e < NE; e += PS_THREADS) wk[e] = (WT)pw[e];
  __syncthreads();

  int n = 0;
  for (int iter = 0; iter < PS_MAX_ITER; ++iter) {
    // arg-max with "smallest linear index wins ties"
    double best = (double)wk[tid];
    int bi = tid;
    for (int e = tid + PS_THREADS; e < NE; e += PS_THREADS) {
      const double v = (double)wk[e];
      if (v > best) { best = v; bi = e; }
    }
    for (int off = 32; off > 0; off >>= 1) {
      const double ov = __shfl_down(best, off);
      const int oi = __shfl_down(bi, off);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { s_val[tid >> 6] = best; s_idx[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
      double b = s_val[0];
      int i = s_idx[0];
      for (int k = 1; k < PS_THREADS / 64; ++k)
        if (s_val[k] > b || (s_val[k] == b && s_idx[k] < i)) { b = s_val[k]; i = s_idx[k]; }
      const int peak_n_id_2 = i / LCS_N_IDX, peak_ind = i % LCS_N_IDX;
      const double peak_pow = b;
      int stop = 0;
      if (peak_pow < z[peak_ind]) stop = 1;
      if (!stop) {
        const int fi = fq[peak_n_id_2 * LCS_N_IDX + peak_ind];
        double best_pow = -INFINITY;
        int best_ind = -1;
        const int ds = geo.ds;
        // for (uint16 t=peak_ind-ds; t<=peak_ind+ds; t++): t wraps to >=65534 when peak_ind<ds
        if (peak_ind - ds >= 0) {
          for (int t = peak_ind - ds; t <= peak_ind + ds; ++t) {
            const int tw = t % LCS_N_IDX;
            const int cc = fi * 3 + peak_n_id_2;
            const float v = sg[((size_t)(cc / geo.cpg) * LCS_N_IDX + tw) * LCS_TG + (cc % geo.cpg)];
            if ((double)v > best_pow) { best_pow = v; best_ind = tw; }
          }
        }
        if (n < LCS_MAXP) {
          lcs_cell c;
          cell_init(c);
          c.fc_requested = params[slot].fc_req;
          c.fc_programmed = params[slot].fc_prog;
          c.pss_pow = peak_pow;
          c.ind = best_ind;
          c.freq = fset[fi];
          c.n_id_2 = peak_n_id_2;
          out[n] = c;
        }
      }
      s_peak_pow = peak_pow;
      s_peak_lin = i;
      s_stop = stop;
    }
    __syncthreads();
    if (s_stop) break;
    ++n;
    const int row = s_peak_lin / LCS_N_IDX, col = s_peak_lin % LCS_N_IDX;
    for (int t = tid; t <= 2 * 274; t += PS_THREADS) {
      const int cc = ((col + t - 274) % LCS_N_IDX + LCS_N_IDX) % LCS_N_IDX;
      wk[row * LCS_N_IDX + cc] = (WT)0;
    }
    __syncthreads();
    const double thresh = s_peak_pow * udb10_m12;
    for (int e = tid; e < NE; e += PS_THREADS)
      if ((double)wk[e] < thresh) wk[e] = (WT)0;
    __syncthreads();
  }
  if (tid == 0) npeaks[slot] = n;
}

// Fused-chain variant: the 3x9600 working copy lives in REGISTERS (113 fp32 per thread of a
// 256-thread workgroup; the collapsed powers are fp32 values, so fp32 compares are the fp64 ones).
// A 4-wave workgroup with no LDS to speak of can be placed next to the resident correlation
// workgroups of the following batch; the 1024-thread version above cannot (it needs 4 free wave
// slots on every SIMD of one CU at once and waited ~6 ms for them in the pipelined chain).
#define PSR_THREADS 256
#define PSR_RPR ((LCS_N_IDX + PSR_THREADS - 1) / PSR_THREADS)   // registers per PSS row (38, last one half padding)
#define PSR_REGS (3 * PSR_RPR)
__global__ __launch_bounds__(PSR_THREADS) __attribute__((amdgpu_waves_per_eu(2, 8))) void k_peak_search_reg(const float *__restrict__ pow32, const int *__restrict__ frq,
                                                                 const double *__restrict__ zth,
                                                                 const float *__restrict__ single,
                                                                 const double *__restrict__ fset,
                                                                 const SlotParams *__restrict__ params,
                                                                 lcs_cell *__restrict__ peaks, int *__restrict__ npeaks,
                                                                 XcGeom geo, double udb10_m12) {
  LCS_TAIL_PRIO();
  const int slot = blockIdx.x;
  const int tid = threadIdx.x;
  const int NE = 3 * LCS_N_IDX;
  const float *pw = pow32 + (size_t)slot * NE;
  const int *fq = frq + (size_t)slot * NE;
  const double *z = zth + (size_t)slot * LCS_N_IDX;
  const float *sg = single + (size_t)slot * geo.G * LCS_N_IDX * LCS_TG;   // group-major (pss_xcorr.hip)
  lcs_cell *out = peaks + (size_t)slot * LCS_MAXP;

  __shared__ float s_val[PSR_THREADS / 64];
  __shared__ int s_idx[PSR_THREADS / 64];
  __shared__ double s_peak_pow;
  __shared__ int s_peak_lin;
  __shared__ int s_stop;

  float v[PSR_REGS];
  // register r of thread tid holds row r / 38, column (r % 38) * 256 + tid: row and column base are
  // compile-time per register, and ascending r is ascending linear index (first-maximum rule)
#pragma unroll
  for (int r = 0; r < PSR_REGS; ++r) {
    const int ec = (r % PSR_RPR) * PSR_THREADS + tid;
    v[r] = (ec < LCS_N_IDX) ? pw[(r / PSR_RPR) * LCS_N_IDX + ec] : -1.0f;   // powers are >= 0: padding never wins
  }

  int n = 0;
  for (int iter = 0; iter < PS_MAX_ITER; ++iter) {
    float best = v[0];
    int br = 0;
#pragma unroll
    for (int r = 1; r < PSR_REGS; ++r)
      if (v[r] > best) { best = v[r]; br = r; }
    asm volatile("" : "+v"(br));   // keep the select chain on the small constants r (else 113 hoisted index registers)
    int bi = (br / PSR_RPR) * LCS_N_IDX + (br % PSR_RPR) * PSR_THREADS + tid;
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_down(best, off);
      const int oi = __shfl_down(bi, off);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { s_val[tid >> 6] = best; s_idx[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
      float b = s_val[0];
      int i = s_idx[0];
      for (int k = 1; k < PSR_THREADS / 64; ++k)
        if (s_val[k] > b || (s_val[k] == b && s_idx[k] < i)) { b = s_val[k]; i = s_idx[k]; }
      const int peak_n_id_2 = i / LCS_N_IDX, peak_ind = i % LCS_N_IDX;
      const double peak_pow = (double)b;
      int stop = 0;
      if (peak_pow < z[peak_ind]) stop = 1;
      if (!stop) {
        const int fi = fq[peak_n_id_2 * LCS_N_IDX + peak_ind];
        double best_pow = -INFINITY;
        int best_ind = -1;
        const int ds = geo.ds;
        // hypotheses split over GPUs (lcs_foe_*): fq holds global indices, this rank's xc_incoherent_single only its
        // own share [foi0, foi0 + n_f): a peak won elsewhere is listed (the greedy loop must be the same everywhere)
        // but not refined here
        const int fl = fi - geo.foi0;
        const bool mine = geo.foi0 >= 0 && fl >= 0 && fl < geo.n_f;
        if (mine && peak_ind - ds >= 0) {   // uint16 wrap of the reference loop variable, quirk Q2
          for (int t = peak_ind - ds; t <= peak_ind + ds; ++t) {
            const int tw = t % LCS_N_IDX;
            const int cc = fl * 3 + peak_n_id_2;
            const float sv = sg[((size_t)(cc / geo.cpg) * LCS_N_IDX + tw) * LCS_TG + (cc % geo.cpg)];
            if ((double)sv > best_pow) { best_pow = sv; best_ind = tw; }
          }
        }
        if (n < LCS_MAXP) {
          lcs_cell c;
          cell_init(c);
          c.fc_requested = params[slot].fc_req;
          c.fc_programmed = params[slot].fc_prog;
          c.pss_pow = peak_pow;
          c.ind = best_ind;
          c.freq = fset[fi];
          c.n_id_2 = peak_n_id_2;
          c.reserved = mine ? 0 : 1;
          out[n] = c;
        }
      }
      s_peak_pow = peak_pow;
      s_peak_lin = i;
      s_stop = stop;
    }
    __syncthreads();
    if (s_stop) break;
    ++n;
    const int row = s_peak_lin / LCS_N_IDX, col = s_peak_lin % LCS_N_IDX;
    const double thresh = s_peak_pow * udb10_m12;
    const int tc = tid - col;
#pragma unroll
    for (int r = 0; r < PSR_REGS; ++r) {
      const int d = (r % PSR_RPR) * PSR_THREADS + tc;   // column - col, in (-9600, 9600)
      const bool real = (r % PSR_RPR != PSR_RPR - 1) || (d + col < LCS_N_IDX);
      const int ad = d < 0 ? -d : d;                     // circular distance <= 274  <=>  |d| <= 274 or |d| >= 9600 - 274
      const bool cancel = (r / PSR_RPR == row) && (ad <= 274 || ad >= LCS_N_IDX - 274);
      if (real && (cancel || (double)v[r] < thresh)) v[r] = 0.0f;
    }
    __syncthreads();   // s_* are rewritten by thread 0 in the next iteration
  }
  if (tid == 0) npeaks[slot] = n;
}

int lcs_launch_peak_search(lcs_ctx *c, int n_buf, const XcGeom &geo, double udb10_m12, bool fp32_exact) {
  if (fp32_exact) {
    hipLaunchKernelGGL(k_peak_search_reg, dim3(n_buf), dim3(PSR_THREADS), 0, c->stream, reinterpret_cast<const float *>(c->work), c->frq, c->zth, c->single,
                       c->fset, c->params, c->peaks, c->npeaks, geo, udb10_m12);
  } else
    hipLaunchKernelGGL(k_peak_search, dim3(n_buf), dim3(PS_THREADS), 0, c->stream, c->pow_, c->frq, c->zth, c->single,
                       c->fset, c->params, c->work, c->peaks, c->npeaks, geo, udb10_m12);
  HIPCHK(c, hipGetLastError());
  return LCS_OK;
}


// ------------------------------------------------------------ hypotheses split over GPUs: packed (pow, frq) words
// word = (bits(pow as float) << 32) | (0xFFFFFFFF - global foi): a MAX all-reduce over ranks is the reference's
// first-maximum rule over the whole frequency axis (lcs_api.hip: lcs_foe_partial).  meta = sp_incoherent[9600], n_comb_xc.
__global__ __launch_bounds__(256) void k_foe_pack(const float *__restrict__ pow32, const int *__restrict__ frq, const double *__restrict__ spinc,
                                                  long long *__restrict__ words, double *__restrict__ meta, XcGeom geo) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 3 * LCS_N_IDX) {
    long long w = -1;                                    // below every real word (pow >= 0 packs non-negative)
    if (geo.foi0 >= 0) w = ((long long)__float_as_uint(pow32[i]) << 32) | (long long)(0xFFFFFFFFu - (unsigned)(frq[i] + geo.foi0));
    words[i] = w;
  }
  if (i < LCS_N_IDX) meta[i] = spinc[i];
  if (i == LCS_N_IDX) meta[i] = (double)geo.n_comb;
}
__global__ __launch_bounds__(256) void k_foe_unpack(const long long *__restrict__ words, const double *__restrict__ meta, double *__restrict__ pow_,
                                                    float *__restrict__ pow32, int *__restrict__ frq, double *__restrict__ spinc, double *__restrict__ zth,
                                                    double R_th1, double rx_cutoff, int ds) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 3 * LCS_N_IDX) {
    const long long w = words[i];
    const float p = __uint_as_float((unsigned)(w >> 32));
    pow32[i] = p;
    pow_[i] = (double)p;
    frq[i] = (int)(0xFFFFFFFFu - (unsigned)(w & 0xFFFFFFFFll));
  }
  if (i < LCS_N_IDX) {
    const double v = meta[i];
    spinc[i] = v;
    zth[i] = R_th1 * v / rx_cutoff / 137 / 2 / (int)meta[LCS_N_IDX] / (2 * ds + 1);      // src/CellSearch.cpp:500-503, as k_sp_fold
  }
}
int lcs_launch_foe_pack(lcs_ctx *c, const XcGeom &geo, long long *d_words, double *d_meta) {
  hipLaunchKernelGGL(k_foe_pack, dim3((3 * LCS_N_IDX + 255) / 256), dim3(256), 0, c->stream, reinterpret_cast<const float *>(c->work), c->frq, c->spinc,
                     d_words, d_meta, geo);
  HIPCHK(c, hipGetLastError());
  return LCS_OK;
}
int lcs_launch_foe_unpack(lcs_ctx *c, const XcGeom &geo, const long long *d_words, const double *d_meta) {
  const double R_th1 = lcs_tables::chi2cdf_inv(1 - pow(10.0, -12), 2.0 * geo.n_comb * (2 * geo.ds + 1));
  const double rx_cutoff = (6 * 12 * 15e3 / 2 + 4 * 15e3) / (30720000.0 / 16 / 2);
  hipLaunchKernelGGL(k_foe_unpack, dim3((3 * LCS_N_IDX + 255) / 256), dim3(256), 0, c->stream, d_words, d_meta, c->pow_, reinterpret_cast<float *>(c->work),
                     c->frq, c->spinc, c->zth, R_th1, rx_cutoff, geo.ds);
  HIPCHK(c, hipGetLastError());
  return LCS_OK;
}
