// mfma_valu_coissue.hip -- which VALU instructions issue in the shadow of v_mfma_i32_16x16x64_i8 on gfx950?  (developer
// micro-benchmark, not product code.)  Every wave runs ITER x 8 independent-accumulator MFMAs with NV independent VALU
// instructions of one kind behind each of them; 2 waves per SIMD (the correlation kernel's occupancy).  Reported: time per
// MFMA per SIMD -- flat in NV = the instruction hides under the matrix pipe, growing = it does not.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// KIND 0: v_cvt_f32_i32   1: v_lshlrev_b32   2: v_fma_f32   3: v_pk_fma_f32   4: v_fmac_f32 (dependent chain per register)
// KIND 5: ds_read2_b32 (8 bytes per lane)   6: ds_read_b128   (LDS reads as the kernel issues them; waited for once per 8 MFMAs)
// KIND 7: ds_read_b64   8: ds_read2_b64 (16 contiguous bytes, 8-byte aligned)   9: ds_read_b128 at an address that is only 8-byte aligned
// KIND 10: ds_read_b128 at an address that is only 4-byte aligned   11: ds_read_b32   12: ds_read_b64 at a 4-byte aligned address
template <int KIND, int NV>
__global__ __launch_bounds__(256) void k(int iters, float *out) {
  __shared__ int lds[256 * 8 + 64];
  for (int e = threadIdx.x; e < 256 * 8 + 64; e += 256) lds[e] = e;
  __syncthreads();
  // conflict-free addresses: consecutive lanes read consecutive 8-byte (ds_read2_b32) / 16-byte (ds_read_b128) pieces
  const unsigned la = (unsigned)(size_t)(lds + threadIdx.x * ((KIND == 5 || KIND == 7 || KIND == 12) ? 2 : (KIND == 11 ? 1 : 4)) + (KIND == 8 || KIND == 9 ? 2 : ((KIND == 10 || KIND == 12) ? 1 : 0)));
  i32x4 a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)blockIdx.x, 8};
  i32x4 acc[8];
  i32x4 lr[8];
  long long l2[8];
  int vi[8];
  float vf[8], vg[8];
  f32x2 vp[8], vq[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { acc[i] = (i32x4){0, 0, 0, 0}; vi[i] = threadIdx.x + i; vf[i] = 0.f; vg[i] = 1.f + i; vp[i] = (f32x2){1.f, 2.f}; vq[i] = (f32x2){.5f, .25f}; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int s = (i * NV + q) & 7;
        if (KIND == 0) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(vf[s]) : "v"(vi[s]));
        else if (KIND == 1) asm volatile("v_lshlrev_b32 %0, 8, %1" : "=v"(vi[s]) : "v"(vi[(s + 1) & 7]));
        else if (KIND == 2) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(vf[s]) : "v"(vg[s]), "v"(vg[(s + 1) & 7]), "v"(vg[(s + 2) & 7]));
        else if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(vp[s]) : "v"(vq[s]), "v"(vq[(s + 1) & 7]), "v"(vq[(s + 2) & 7]));
        else if (KIND == 4) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(vf[s]) : "v"(vg[s]), "v"(vg[(s + 1) & 7]));
        else if (KIND == 5) asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1" : "=v"(l2[s]) : "v"(la));
        else if (KIND == 7 || KIND == 12) asm volatile("ds_read_b64 %0, %1" : "=v"(l2[s]) : "v"(la));
        else if (KIND == 11) asm volatile("ds_read_b32 %0, %1" : "=v"(vi[s]) : "v"(la));
        else if (KIND == 8) asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:1" : "=v"(lr[s]) : "v"(la));
        else asm volatile("ds_read_b128 %0, %1" : "=v"(lr[s]) : "v"(la));
      }
    }
    if (KIND >= 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float r = 0;
  if (KIND >= 5) for (int i = 0; i < 8; ++i) r += lr[i][0] + (float)l2[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3] + vf[i] + vi[i] + vp[i][0] + vp[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND, int NV>
double run() {
  float *out;
  (void)hipMalloc(&out, 256 * 4096 * sizeof(float));
  const int iters = 20000, wg_per_cu = 2, grid = 256 * wg_per_cu;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, NV>), dim3(grid), dim3(256), 0, 0, 2000, out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, NV>), dim3(grid), dim3(256), 0, 0, iters, out);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipFree(out);
  return ms * 1e6 / ((double)iters * 8 * wg_per_cu);
}

template <int KIND>
void row(const char *name) {
  printf("%-16s ns per MFMA per SIMD with 0..6 of them behind every MFMA: %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f\n", name, run<KIND, 0>(),
         run<KIND, 1>(), run<KIND, 2>(), run<KIND, 3>(), run<KIND, 4>(), run<KIND, 5>(), run<KIND, 6>());
}

int main() {
  row<0>("v_cvt_f32_i32");
  row<1>("v_lshlrev_b32");
  row<2>("v_fma_f32");
  row<3>("v_pk_fma_f32");
  row<4>("v_fmac_f32");
  row<5>("ds_read2_b32");
  row<6>("ds_read_b128");
  row<7>("ds_read_b64");
  row<11>("ds_read_b32");
  row<12>("ds_read_b64 a4");
  return 0;
}
