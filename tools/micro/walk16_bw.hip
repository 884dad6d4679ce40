// Memory-side skeletons of kpe_bs (no arithmetic): a lane loads and stores 16 bytes per row step for
// two strips of R rows (two loads in flight per strip), 10 000 frames of 320 x 240 bytes.  What does the
// mapping of a wave's 64 lanes onto memory cost?
//   frames62 : 62 chunks of one image row laid over the frames back to back + 2 halo lanes   (kpe_bs round 3a)
//   frames60 : three whole rows (same y) of three consecutive frames, 4 lanes idle
//   rows3    : three consecutive rows of ONE frame (960 contiguous bytes), a step is three rows
//   lin64    : 1 KB contiguous and aligned per wave access (the ceiling for a walking wave)
// LDS bytes per workgroup are a launch parameter: they set the workgroups per CU (occupancy).
// hipcc --offload-arch=gfx950 -O3 tools/micro/walk16_bw.hip -o /tmp/walk16_bw && /tmp/walk16_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int W = 320, H = 240, WC = W / 16;
typedef unsigned int u32x4 __attribute__((__vector_size__(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k_walk(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t n, int R, uint32_t groups) {
  extern __shared__ uint32_t s_pad[];
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const uint32_t g = blockIdx.x * 4 + wv;
  if (g >= groups) return;
  const int ya = blockIdx.y * 2 * R, yb = ya + R;
  size_t base;       // byte offset of the lane's chunk in row 0 of its frame
  bool ld_ok = true, st_ok = true;
  int rows_per_step = 1;
  if (MODE == 0) {
    const long long V = (long long)n * WC, L = (long long)g * 62 - 1 + lane;
    const long long Lc = L < 0 ? 0 : (L >= V ? V - 1 : L);
    st_ok = L >= 0 && L < V && lane >= 1 && lane <= 62;
    base = (size_t)(Lc / WC) * W * H + (size_t)(Lc % WC) * 16;
  } else if (MODE == 1) {
    const uint32_t f = g * 3 + lane / 20;
    ld_ok = st_ok = lane < 60 && f < n;
    base = (size_t)(f < n ? f : n - 1) * W * H + (size_t)(lane % 20) * 16;
  } else if (MODE == 2) {
    ld_ok = st_ok = lane < 60;
    base = (size_t)g * W * H + (size_t)(lane < 60 ? lane : 0) * 16;  // rows r, r+1, r+2 are contiguous
    rows_per_step = 3;
  } else {
    base = (size_t)g * W * H + (size_t)lane * 16;
  }
  const uint8_t* src = in + base;
  uint8_t* dst = out + base;
  if (MODE == 3) {  // bytes [ya * W, (ya + 2 R) * W) of the frame in 1-KB steps
    const int steps = 2 * R * W / 1024;
    const size_t o = (size_t)ya * W;
    u32x4 p0 = *(const u32x4*)(src + o), p1 = *(const u32x4*)(src + o + 1024);
    for (int j = 0; j < steps; j += 2) {
      u32x4 v0 = p0, v1 = p1;
      if (j + 2 < steps) p0 = *(const u32x4*)(src + o + (size_t)(j + 2) * 1024);
      if (j + 3 < steps) p1 = *(const u32x4*)(src + o + (size_t)(j + 3) * 1024);
      v0[0] ^= 1u;
      v1[0] ^= 1u;
      *(u32x4*)(dst + o + (size_t)j * 1024) = v0;
      if (j + 1 < steps) *(u32x4*)(dst + o + (size_t)(j + 1) * 1024) = v1;
    }
    return;
  }
  const int steps = R / rows_per_step;
  auto ld = [&](int y) -> u32x4 {
    u32x4 z = {0, 0, 0, 0};
    return ld_ok ? *(const u32x4*)(src + (size_t)(y < H ? y : H - 1) * W) : z;
  };
  const int rs = rows_per_step;
  u32x4 a0 = ld(ya), b0 = ld(yb), a1 = ld(ya + rs), b1 = ld(yb + rs);
  for (int j = 0; j < steps; j += 2) {
    u32x4 va = a0, vb = b0;
    if (j + 2 < steps) a0 = ld(ya + (j + 2) * rs), b0 = ld(yb + (j + 2) * rs);
    va[0] ^= 1u;
    vb[0] ^= 1u;
    if (st_ok) {
      *(u32x4*)(dst + (size_t)(ya + j * rs) * W) = va;
      *(u32x4*)(dst + (size_t)(yb + j * rs) * W) = vb;
    }
    if (j + 1 < steps) {
      va = a1, vb = b1;
      if (j + 3 < steps) a1 = ld(ya + (j + 3) * rs), b1 = ld(yb + (j + 3) * rs);
      va[0] ^= 1u;
      vb[0] ^= 1u;
      if (st_ok) {
        *(u32x4*)(dst + (size_t)(ya + (j + 1) * rs) * W) = va;
        *(u32x4*)(dst + (size_t)(yb + (j + 1) * rs) * W) = vb;
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_linear(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n16) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) out[i] = in[i];
}

int main(int argc, char** argv) {
  const uint32_t n = argc > 1 ? atoi(argv[1]) : 10000;
  const size_t bytes = (size_t)n * W * H;
  uint8_t *in, *out;
  (void)hipMalloc(&in, bytes + 8192);
  (void)hipMalloc(&out, bytes + 8192);
  (void)hipMemset(in, 3, bytes);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  auto time = [&](const char* name, int R, int lds, auto&& launch) {
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
    printf("%-10s R=%-3d lds=%-6d avg %.3f ms  best %.3f ms  %.2f TB/s (read + write)%s\n", name, R, lds, sum / 10, best,
           2.0 * bytes / (sum / 10) / 1e9, e == hipSuccess ? "" : hipGetErrorString(e));
    fflush(stdout);
  };
  time("linear", 0, 0, [&] { hipLaunchKernelGGL(k_linear, dim3((uint32_t)((bytes / 16 + 255) / 256)), dim3(256), 0, 0, (const uint4*)in, (uint4*)out, bytes / 16); });
  for (int lds : {0, 32 * 1024, 40 * 1024})
    for (int R : {12, 24, 60}) {
      const uint32_t g0 = (uint32_t)(((size_t)n * WC + 61) / 62), g1 = (n + 2) / 3, g2 = n;
      const uint32_t gy = H / (2 * R);
      time("frames62", R, lds, [&] { hipLaunchKernelGGL(k_walk<0>, dim3((g0 + 3) / 4, gy), dim3(256), lds, 0, in, out, n, R, g0); });
      time("frames60", R, lds, [&] { hipLaunchKernelGGL(k_walk<1>, dim3((g1 + 3) / 4, gy), dim3(256), lds, 0, in, out, n, R, g1); });
      time("rows3", R, lds, [&] { hipLaunchKernelGGL(k_walk<2>, dim3((g2 + 3) / 4, gy), dim3(256), lds, 0, in, out, n, R, g2); });
      if ((2 * R * W) % 1024 == 0 || R == 12) {
        const int R3 = R == 12 ? 8 : R;  // 16 rows = 5 KB
        time("lin64", R3, lds, [&] { hipLaunchKernelGGL(k_walk<3>, dim3((g2 + 3) / 4, H / (2 * R3)), dim3(256), lds, 0, in, out, n, R3, g2); });
      }
    }
  return 0;
}
