// mfma_rate.hip -- issue rate of the int8 MFMA shapes on gfx950 (developer micro-benchmark, not product code).
// Every wave runs ITER x NACC independent-accumulator MFMAs; reports shader cycles per MFMA per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

template <int SHAPE, int NACC>
__global__ __launch_bounds__(256) void k(int iters, int *out, long long *cyc) {
  i32x4 a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)blockIdx.x, 8};
  long long t0 = 0, t1 = 0;
  if constexpr (SHAPE == 0) {          // 16x16x64 i8
    i32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (i32x4){0, 0, 0, 0};
    t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
    }
    t1 = clock64();
    int s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  } else if constexpr (SHAPE == 1) {   // 32x32x32 i8
    i32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
    }
    t1 = clock64();
    int s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  } else {                             // 16x16x32 i8 (legacy shape, 8-byte operands)
    long a8 = threadIdx.x * 0x0101010101010101L, b8 = 0x0203040506070809L + blockIdx.x;
    i32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (i32x4){0, 0, 0, 0};
    t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_i32_16x16x32_i8 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a8), "v"(b8));
    }
    t1 = clock64();
    int s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int SHAPE, int NACC>
void run(const char *name, int wg_per_cu, double ops_per_mfma) {
  int *out; long long *cyc;
  hipMalloc(&out, 256 * 4096 * sizeof(int));
  hipMalloc(&cyc, 8);
  const int iters = 4000, grid = 256 * wg_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(grid), dim3(256), 0, 0, 100, out, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(grid), dim3(256), 0, 0, iters, out, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n_per_wave = (double)iters * NACC;
  const double total = n_per_wave * 4 * grid;
  printf("%-28s waves/SIMD %d  clock64 ticks per MFMA per wave %.2f  wall: %.3f ms  %.1f TOP/s  => %.2f cycles@2.4GHz per MFMA per SIMD\n", name,
         wg_per_cu, (double)c / n_per_wave, ms, total * ops_per_mfma / (ms * 1e-3) / 1e12,
         (ms * 1e-3) * 2.4e9 / (n_per_wave * wg_per_cu));
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0, 8>("i32_16x16x64_i8 x8acc", 1, 32768.0);
  run<0, 8>("i32_16x16x64_i8 x8acc", 2, 32768.0);
  run<0, 16>("i32_16x16x64_i8 x16acc", 1, 32768.0);
  run<1, 4>("i32_32x32x32_i8 x4acc", 1, 65536.0);
  run<1, 4>("i32_32x32x32_i8 x4acc", 2, 65536.0);
  run<1, 8>("i32_32x32x32_i8 x8acc", 1, 65536.0);
  run<2, 8>("i32_16x16x32_i8 x8acc", 1, 16384.0);
  run<2, 8>("i32_16x16x32_i8 x8acc", 2, 16384.0);
  return 0;
}
