// Why does a wave that walks down memory (load 1 KB, store 1 KB, next KB ...) stop at ~5 TB/s when a
// one-shot copy (one load and one store per wave) reaches 6.1?  gfx9 has ONE counter (vmcnt) for loads and
// stores, decremented in issue order: a wait for load data also waits for every older store's ack.
//   walk     : per step wait(load j) -> store j -> issue load j+D      (kpe_bs's loop, D loads in flight)
//   loadonly : the same loop without the stores (one store at the end)
//   storeonly: the same loop without the loads
//   burst    : B loads issued, then B stores of the previous burst: a load wait never covers a store
//              younger than one burst
// 10 000 frames of 320 x 240 bytes; a wave owns `span` KB of consecutive memory.
// hipcc --offload-arch=gfx950 -O3 tools/micro/walk_split_bw.hip -o /tmp/walk_split_bw && /tmp/walk_split_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((__vector_size__(16)));

template <int KIND, int D>
__global__ __launch_bounds__(256) void k_walk(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t waves, int span) {
  const uint32_t lane = threadIdx.x & 63u, g = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= waves) return;
  const u32x4* src = (const u32x4*)(in + (size_t)g * span * 1024) + lane;
  u32x4* dst = (u32x4*)(out + (size_t)g * span * 1024) + lane;
  u32x4 q[D];
  u32x4 acc = {lane, 1, 2, 3};
  if (KIND != 2)
#pragma unroll
    for (int d = 0; d < D; ++d) q[d] = src[(size_t)d * 64];
  for (int j = 0; j < span; j += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      u32x4 v = KIND == 2 ? acc : q[d];
      if (KIND != 2 && j + d + D < span) q[d] = src[(size_t)(j + d + D) * 64];
      v[0] ^= 1u;
      if (KIND == 1) acc ^= v;
      else dst[(size_t)(j + d) * 64] = v;
    }
  }
  if (KIND == 1) dst[0] = acc;
}

template <int B>
__global__ __launch_bounds__(256) void k_burst(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t waves, int span) {
  const uint32_t lane = threadIdx.x & 63u, g = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= waves) return;
  const u32x4* src = (const u32x4*)(in + (size_t)g * span * 1024) + lane;
  u32x4* dst = (u32x4*)(out + (size_t)g * span * 1024) + lane;
  u32x4 a[B], b[B];
#pragma unroll
  for (int d = 0; d < B; ++d) a[d] = src[(size_t)d * 64];
  for (int j = 0; j < span; j += B) {
    if (j + B < span)
#pragma unroll
      for (int d = 0; d < B; ++d) b[d] = src[(size_t)(j + B + d) * 64];  // next burst first ...
#pragma unroll
    for (int d = 0; d < B; ++d) {  // ... then this burst's stores
      a[d][0] ^= 1u;
      dst[(size_t)(j + d) * 64] = a[d];
    }
#pragma unroll
    for (int d = 0; d < B; ++d) a[d] = b[d];
  }
}

__global__ __launch_bounds__(256) void k_linear(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n16) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) out[i] = in[i];
}

int main(int argc, char** argv) {
  const size_t bytes = (size_t)10000 * 320 * 240;
  uint8_t *in, *out;
  (void)hipMalloc(&in, bytes + 8192);
  (void)hipMalloc(&out, bytes + 8192);
  (void)hipMemset(in, 3, bytes);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  auto time = [&](const char* name, int span, double moved, auto&& launch) {
    launch();
    (void)hipDeviceSynchronize();
    float best = 1e9f, sum = 0;
    for (int it = 0; it < 10; ++it) {
      (void)hipEventRecord(a);
      launch();
      (void)hipEventRecord(b);
      (void)hipEventSynchronize(b);
      float ms;
      (void)hipEventElapsedTime(&ms, a, b);
      best = ms < best ? ms : best;
      sum += ms;
    }
    hipError_t e = hipGetLastError();
    printf("%-12s span=%-3d KB  avg %.3f ms  best %.3f ms  %.2f TB/s%s\n", name, span, sum / 10, best, moved / (sum / 10) / 1e9,
           e == hipSuccess ? "" : hipGetErrorString(e));
    fflush(stdout);
  };
  time("linear", 1, 2.0 * bytes, [&] { hipLaunchKernelGGL(k_linear, dim3((uint32_t)((bytes / 16 + 255) / 256)), dim3(256), 0, 0, (const uint4*)in, (uint4*)out, bytes / 16); });
  for (int span : {12, 24, 60}) {
    const uint32_t waves = (uint32_t)(bytes / 1024 / span);
    const dim3 grid((waves + 3) / 4), blk(256);
#define RUN(name, kern, moved) time(name, span, moved, [&] { hipLaunchKernelGGL(kern, grid, blk, 0, 0, in, out, waves, span); })
    RUN("walk d2", (k_walk<0, 2>), 2.0 * bytes);
    RUN("walk d4", (k_walk<0, 4>), 2.0 * bytes);
    RUN("walk d6", (k_walk<0, 6>), 2.0 * bytes);
    RUN("loadonly d2", (k_walk<1, 2>), 1.0 * bytes);
    RUN("loadonly d4", (k_walk<1, 4>), 1.0 * bytes);
    RUN("storeonly", (k_walk<2, 2>), 1.0 * bytes);
    RUN("burst 2", (k_burst<2>), 2.0 * bytes);
    RUN("burst 4", (k_burst<4>), 2.0 * bytes);
    RUN("burst 6", (k_burst<6>), 2.0 * bytes);
  }
  return 0;
}
