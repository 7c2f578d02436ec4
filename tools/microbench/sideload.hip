// sideload.hip -- synthetic side loads for tools/microbench/corun.py: what kind of co-running work slows the PSS
// correlation kernel?  Every load runs on its own stream, optionally confined to the first `n_cu` CUs
// (hipExtStreamCreateWithCUMask), so that slot competition on the correlation's CUs can be told apart from pressure on
// what the whole chip shares (L2 / fabric / HBM, the workgroup dispatcher).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libsideload.so sideload.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void k_alu64(double *out, int iters) {           // fp64 ALU only: a few registers, no memory traffic
  double x = threadIdx.x * 1e-3, y = 1.000001;
  for (int i = 0; i < iters; ++i) { x = x * y + 0.5; y = y * 0.999999 + 1e-7; }
  if (x == 12345.678) out[0] = x + y;
}
__global__ void k_stream_read(const uint4 *__restrict__ src, size_t n, uint4 *out) {      // HBM / L2 streaming reads
  uint4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = src[i];
    acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
  }
  if (acc.x == 0x12345678u) out[0] = acc;
}
__global__ void k_stream_write(uint4 *__restrict__ dst, size_t n) {
  const uint4 v = {1, 2, 3, 4};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}
__global__ void k_empty(int *p) { if (p && threadIdx.x == 99999) p[0] = 1; }
// many registers + LDS: a workgroup that cannot share a CU slot with two resident correlation workgroups
__global__ __launch_bounds__(64) void k_fat(double *out, int iters) {
  __shared__ double lds[4096];
  double r[48];
  for (int j = 0; j < 48; ++j) r[j] = threadIdx.x + j;
  for (int i = 0; i < iters; ++i) {
    for (int j = 0; j < 48; ++j) r[j] = r[j] * 1.0000001 + 0.25;
    lds[(threadIdx.x + i) & 4095] = r[i % 48];
  }
  double s = 0;
  for (int j = 0; j < 48; ++j) s += r[j];
  if (s == 1.2345) out[0] = s + lds[5];
}

extern "C" {
hipStream_t sl_stream(int n_cu) {       // n_cu <= 0: ordinary stream
  hipStream_t s = nullptr;
  if (n_cu <= 0) { (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking); return s; }
  uint32_t m[8] = {0};
  for (int b = 0; b < n_cu && b < 256; ++b) m[b >> 5] |= 1u << (b & 31);
  if (hipExtStreamCreateWithCUMask(&s, 8, m) != hipSuccess) return nullptr;
  return s;
}
void *sl_alloc(size_t bytes) { void *p = nullptr; (void)hipMalloc(&p, bytes); (void)hipMemset(p, 1, bytes); return p; }
void sl_alu64(hipStream_t s, int blocks, int iters, void *scratch) { hipLaunchKernelGGL(k_alu64, dim3(blocks), dim3(64), 0, s, (double *)scratch, iters); }
void sl_fat(hipStream_t s, int blocks, int iters, void *scratch) { hipLaunchKernelGGL(k_fat, dim3(blocks), dim3(64), 0, s, (double *)scratch, iters); }
void sl_read(hipStream_t s, int blocks, void *src, size_t bytes, void *scratch) { hipLaunchKernelGGL(k_stream_read, dim3(blocks), dim3(256), 0, s, (const uint4 *)src, bytes / 16, (uint4 *)scratch); }
void sl_write(hipStream_t s, int blocks, void *dst, size_t bytes) { hipLaunchKernelGGL(k_stream_write, dim3(blocks), dim3(256), 0, s, (uint4 *)dst, bytes / 16); }
void sl_empty(hipStream_t s, int blocks, int launches) { for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(64), 0, s, (int *)nullptr); }
void sl_sync(hipStream_t s) { (void)hipStreamSynchronize(s); }
}
