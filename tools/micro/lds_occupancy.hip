// How many 256-thread workgroups with S KB of static LDS does a gfx950 CU run side by side?
// Each workgroup spins ~200 us; 256 CUs x R workgroups take ceil(R / resident) x 200 us.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KB>
__global__ __launch_bounds__(256) void spin(unsigned* out, long long cycles) {
  __shared__ unsigned buf[KB * 256];
  buf[threadIdx.x] = threadIdx.x;
  __syncthreads();
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) out[blockIdx.x] = buf[(blockIdx.x * 7) % (KB * 256)];
}
template <int KB>
void run(unsigned* d) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int R = 16;
  hipLaunchKernelGGL(spin<KB>, dim3(256 * R), dim3(256), 0, 0, d, 20000LL);  // 100 MHz clock: 200 us
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(spin<KB>, dim3(256 * R), dim3(256), 0, 0, d, 20000LL);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("LDS %3d KB per workgroup: %.2f ms for %d workgroups per CU -> ~%.1f resident per CU\n", KB, ms, R, R * 0.2 / ms);
}
int main() {
  unsigned* d; hipMalloc(&d, 1 << 20);
  run<8>(d); run<16>(d); run<20>(d); run<32>(d); run<35>(d); run<40>(d); run<64>(d); run<70>(d); run<80>(d);
  return 0;
}
