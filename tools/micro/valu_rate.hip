// VALU issue rate of a CU: wave-instructions per cycle per SIMD with W waves per SIMD resident.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void spin(unsigned* out, int iters) {
  unsigned a = threadIdx.x, b = a * 3 + 1, c = a ^ 0x55, d = a + 7, e = b + 9, f = c + 11, g = d + 13, h = e + 15;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {  // 8 independent chains of 3-input ops
      asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
      asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(b) : "v"(c), "v"(d));
      asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(c) : "v"(d), "v"(e));
      asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(d) : "v"(e), "v"(f));
      asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(e) : "v"(f), "v"(g));
      asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(f) : "v"(g), "v"(h));
      asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(g) : "v"(h), "v"(a));
      asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(h) : "v"(a), "v"(b));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}
int main() {
  unsigned* d; (void)hipMalloc(&d, 256 * 2048 * 4 * 2);
  hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
  printf("clock %d kHz, CUs %d\n", pr.clockRate, pr.multiProcessorCount);
  for (int threads : {256, 512, 1024, 2048}) {  // per CU: 1, 2, 4, 8 waves per SIMD
    int bs = threads > 1024 ? 1024 : threads, blocks = 256 * (threads / bs);
    int iters = 2000;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(spin, dim3(blocks), dim3(bs), 0, 0, d, 10);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(spin, dim3(blocks), dim3(bs), 0, 0, d, iters);
    (void)hipEventRecord(b); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double winstr = (double)blocks * bs / 64 * iters * 128;
    printf("%4d threads/CU (%d waves/SIMD): %.3f ms, %.3f wave-instr per ns chip-wide = %.3f per cycle per SIMD at 2.4 GHz\n",
           threads, threads / 256, ms, winstr / (ms * 1e6), winstr / (ms * 1e6) / 1024 / 2.4);
  }
  return 0;
}
