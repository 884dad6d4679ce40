// Device copy ceilings on this box: bytes per lane per access x rows in flight, 768 MB in + 768 MB out.
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T, int UNROLL>
__global__ __launch_bounds__(256) void copyk(const T* __restrict__ src, T* __restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
  T v[UNROLL];
#pragma unroll
  for (int k = 0; k < UNROLL; ++k) v[k] = i + (size_t)k * 256 < n ? src[i + (size_t)k * 256] : T{};
#pragma unroll
  for (int k = 0; k < UNROLL; ++k)
    if (i + (size_t)k * 256 < n) dst[i + (size_t)k * 256] = v[k];
}
template <typename T, int UNROLL>
void run(const void* s, void* d, size_t bytes, const char* name) {
  size_t n = bytes / sizeof(T);
  dim3 grid((unsigned)((n + 256 * UNROLL - 1) / (256 * UNROLL)));
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL((copyk<T, UNROLL>), grid, dim3(256), 0, 0, (const T*)s, (T*)d, n);
  (void)hipEventRecord(a);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((copyk<T, UNROLL>), grid, dim3(256), 0, 0, (const T*)s, (T*)d, n);
  (void)hipEventRecord(b);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  printf("%-28s %.3f ms per copy, %.0f GB/s (read + write)\n", name, ms / 10, 2.0 * bytes / (ms / 10 * 1e-3) / 1e9);
}
int main() {
  size_t bytes = 768000000;
  void *s, *d;
  (void)hipMalloc(&s, bytes); (void)hipMalloc(&d, bytes);
  (void)hipMemset(s, 1, bytes);
  run<unsigned, 1>(s, d, bytes, "4 B/lane x1");
  run<unsigned, 4>(s, d, bytes, "4 B/lane x4 in flight");
  run<unsigned, 8>(s, d, bytes, "4 B/lane x8 in flight");
  run<uint2, 4>(s, d, bytes, "8 B/lane x4 in flight");
  run<uint4, 1>(s, d, bytes, "16 B/lane x1");
  run<uint4, 2>(s, d, bytes, "16 B/lane x2 in flight");
  run<uint4, 4>(s, d, bytes, "16 B/lane x4 in flight");
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  (void)hipMemcpy(d, s, bytes, hipMemcpyDeviceToDevice);
  (void)hipEventRecord(a);
  for (int r = 0; r < 10; ++r) (void)hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0);
  (void)hipEventRecord(b); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  printf("%-28s %.3f ms per copy, %.0f GB/s\n", "hipMemcpy D2D", ms / 10, 2.0 * bytes / (ms / 10 * 1e-3) / 1e9);
  return 0;
}
