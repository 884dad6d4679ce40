// Memory-side skeletons for a 3x3/5x5 stencil over a batch of small frames (320x240 u8), no
// arithmetic: which access structure reaches the device's copy rate?
//   linear   : 16 B per lane, block b copies bytes [4 KB * b, 4 KB * (b + 1))            (the ceiling)
//   rowwalk  : a lane owns a dword column and walks down R rows, 6 loads in flight        (kpe_main's pattern)
//   tile     : a block of FPB frames x R rows: (R + 2*HALO) rows of every frame loaded at once with
//              16 B per lane into LDS, one LDS read + one LDS write per lane-row, the R output rows
//              stored with 16 B per lane
//   tile_st4 : as tile, but the output rows stored with 4 B per lane, row by row
// hipcc --offload-arch=gfx950 -O3 tools/micro/tile_bw.hip -o /tmp/tile_bw && /tmp/tile_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int W = 320, H = 240, WD = W / 4;

__global__ __launch_bounds__(256) void k_linear(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n16) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) out[i] = in[i];
}

// grid: x = ceil(n * WD / 256) column groups, y = H / R strips
__global__ __launch_bounds__(256) void k_rowwalk(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n, int R) {
  const size_t L = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (L >= (size_t)n * WD) return;
  const uint32_t f = (uint32_t)(L / WD), c = (uint32_t)(L % WD);
  const uint32_t* src = in + (size_t)f * WD * H + c;
  uint32_t* dst = out + (size_t)f * WD * H + c;
  const int y0 = blockIdx.y * R, y1 = min(H, y0 + R);
  uint32_t pf[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) pf[k] = src[(size_t)min(y0 + k, y1 - 1) * WD];
  for (int y = y0; y < y1; y += 6) {
#pragma unroll
    for (int k = 0; k < 6; ++k)
      if (y + k < y1) {
        uint32_t v = pf[k];
        pf[k] = src[(size_t)min(y + k + 6, y1 - 1) * WD];
        dst[(size_t)(y + k) * WD] = v ^ 0x01010101u;
      }
  }
}

// rowwalk with kpe_main's extras, one at a time: HALO = a wave covers 62 columns + 2 halo lanes
// (unaligned 248-B segments), CAND = one more dword per lane and 8 rows into a bit plane,
// SH = two extra rows loaded per strip
template <bool HALO, bool CAND, bool SH>
__global__ __launch_bounds__(256) void k_rowwalk2(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t* __restrict__ cand,
                                                  uint32_t n, int R) {
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const size_t g = (size_t)blockIdx.x * 4 + wv;
  const long long L = HALO ? (long long)g * 62 - 1 + lane : (long long)g * 64 + lane;
  const long long V = (long long)n * WD;
  if ((HALO ? (long long)g * 62 : (long long)g * 64) >= V) return;
  const bool ok = L >= 0 && L < V && (!HALO || (lane >= 1 && lane <= 62));
  const long long Lc = L < 0 ? 0 : (L >= V ? V - 1 : L);
  const uint32_t f = (uint32_t)(Lc / WD), c = (uint32_t)(Lc % WD);
  const uint32_t* src = in + (size_t)f * WD * H + c;
  uint32_t* dst = out + (size_t)f * WD * H + c;
  const int y0 = blockIdx.y * R, y1 = min(H, y0 + R);
  const int ya = SH ? max(0, y0 - 1) : y0, yb = SH ? min(H, y1 + 1) : y1;
  uint32_t pf[6];
  uint32_t acc = 0;
  if (SH) acc = src[(size_t)ya * WD] + src[(size_t)(yb - 1) * WD];
#pragma unroll
  for (int k = 0; k < 6; ++k) pf[k] = src[(size_t)min(y0 + k, y1 - 1) * WD];
  for (int y = y0; y < y1; y += 6) {
#pragma unroll
    for (int k = 0; k < 6; ++k)
      if (y + k < y1) {
        uint32_t v = pf[k];
        pf[k] = src[(size_t)min(y + k + 6, y1 - 1) * WD];
        acc = (acc >> 1) ^ v;
        if (ok) dst[(size_t)(y + k) * WD] = v ^ 0x01010101u;
        if (CAND && ((y + k) & 7) == 7 && ok) cand[((size_t)f * (H / 8) + ((y + k) >> 3)) * WD + c] = acc;
      }
  }
  if (!CAND && acc == 0x12345u && ok) cand[0] = acc;
}

// block = FPB frames x R rows; NT threads = FPB * WD lanes (one per dword column of the tile row)
template <int FPB, int R, int HALO, bool ST4, bool XCD>
__global__ __launch_bounds__(FPB* WD) void k_tile(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t n) {
  constexpr int NT = FPB * WD, RIN = R + 2 * HALO;
  __shared__ __attribute__((aligned(16))) uint8_t s_in[FPB][RIN * W];
  __shared__ __attribute__((aligned(16))) uint8_t s_out[ST4 ? 1 : FPB][ST4 ? 16 : R * W];
  const int strips = H / R;
  uint32_t b = blockIdx.x;
  if (XCD) {  // blocks that share an XCD (b % 8) walk a contiguous range of tiles
    const uint32_t nb = gridDim.x, bq = nb >> 3, br = nb & 7u, x = b & 7u;
    b = (x < br ? x * (bq + 1) : br * (bq + 1) + (x - br) * bq) + (b >> 3);
  }
  const uint32_t fg = b / strips, s = b % strips;
  const int y0 = s * R;
  const int ya = max(0, y0 - HALO), yb = min(H, y0 + R + HALO);  // rows loaded
  const int tid = threadIdx.x;
  // ---- load: every frame's piece is contiguous ((yb - ya) * W bytes) ----
  const int piece16 = (yb - ya) * W / 16;
  for (int j = 0; j < FPB; ++j) {
    const uint32_t f = fg * FPB + j;
    if (f >= n) break;
    const uint4* src = reinterpret_cast<const uint4*>(in + ((size_t)f * H + ya) * W);
    uint4* dst = reinterpret_cast<uint4*>(&s_in[j][(ya - (y0 - HALO)) * W]);
    for (int i = tid; i < piece16; i += NT) {
      uint4 v = src[i];
      v.x ^= 0x01010101u;
      dst[i] = v;
    }
  }
  __syncthreads();
  // ---- "compute": one LDS read and one LDS write (or one 4-byte global store) per lane-row ----
  const int j = tid / WD, c = tid % WD;
  const uint32_t f = fg * FPB + j;
  if (f < n) {
    for (int r = 0; r < R; ++r) {
      uint32_t v = *reinterpret_cast<const uint32_t*>(&s_in[j][(r + HALO) * W + 4 * c]);
      if (ST4)
        *reinterpret_cast<uint32_t*>(out + ((size_t)f * H + y0 + r) * W + 4 * c) = v;
      else
        *reinterpret_cast<uint32_t*>(&s_out[j][r * W + 4 * c]) = v;
    }
  }
  if (!ST4) {
    __syncthreads();
    for (int jj = 0; jj < FPB; ++jj) {
      const uint32_t ff = fg * FPB + jj;
      if (ff >= n) break;
      uint4* dst = reinterpret_cast<uint4*>(out + ((size_t)ff * H + y0) * W);
      const uint4* src = reinterpret_cast<const uint4*>(&s_out[jj][0]);
      for (int i = tid; i < R * W / 16; i += NT) dst[i] = src[i];
    }
  }
}

int main(int argc, char** argv) {
  const uint32_t n = argc > 1 ? atoi(argv[1]) : 10000;
  const size_t bytes = (size_t)n * W * H;
  uint8_t *in, *out;
  (void)hipMalloc(&in, bytes + 4096);
  (void)hipMalloc(&out, bytes + 4096);
  (void)hipMemset(in, 3, bytes);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  auto time = [&](const char* name, auto&& launch) {
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
    printf("%-34s avg %.3f ms  best %.3f ms  %.2f TB/s (read + write)%s\n", name, sum / 10, best, 2.0 * bytes / (sum / 10) / 1e9,
           e == hipSuccess ? "" : hipGetErrorString(e));
    fflush(stdout);
  };
  time("linear 16 B/lane", [&] { hipLaunchKernelGGL(k_linear, dim3((uint32_t)((bytes / 16 + 255) / 256)), dim3(256), 0, 0, (const uint4*)in, (uint4*)out, bytes / 16); });
  for (int R : {24, 48, 240})
    time(R == 24 ? "rowwalk R=24" : R == 48 ? "rowwalk R=48" : "rowwalk R=240",
         [&] { hipLaunchKernelGGL(k_rowwalk, dim3((uint32_t)(((size_t)n * WD + 255) / 256), H / R), dim3(256), 0, 0, (const uint32_t*)in, (uint32_t*)out, n, R); });
  uint32_t* cand;
  (void)hipMalloc(&cand, bytes / 8 + 4096);
#define RW2(HALO, CAND, SH)                                                                                              \
  time("rowwalk2 R=48 halo62=" #HALO " cand=" #CAND " striphalo=" #SH, [&] {                                              \
    const size_t groups = ((size_t)n * WD + (HALO ? 61 : 63)) / (HALO ? 62 : 64);                                         \
    hipLaunchKernelGGL((k_rowwalk2<HALO, CAND, SH>), dim3((uint32_t)((groups + 3) / 4), H / 48), dim3(256), 0, 0,          \
                       (const uint32_t*)in, (uint32_t*)out, cand, n, 48);                                                 \
  });
  RW2(false, false, false)
  RW2(true, false, false)
  RW2(false, true, false)
  RW2(false, false, true)
  RW2(true, true, true)
#define TILE(FPB, R, HALO, ST4, XCD)                                                                                     \
  time("tile FPB=" #FPB " R=" #R " halo=" #HALO " st4=" #ST4 " xcd=" #XCD,                                               \
       [&] { hipLaunchKernelGGL((k_tile<FPB, R, HALO, ST4, XCD>), dim3((n + FPB - 1) / FPB * (H / R)), dim3(FPB * WD), 0, 0, in, out, n); });
  TILE(4, 8, 0, false, false)
  TILE(4, 16, 0, false, false)
  TILE(4, 8, 2, false, false)
  TILE(4, 16, 2, false, false)
  TILE(4, 16, 2, false, true)
  TILE(4, 24, 2, false, true)
  TILE(4, 16, 2, true, true)
  TILE(4, 8, 2, true, true)
  TILE(2, 16, 2, false, true)
  TILE(2, 24, 2, false, true)
  TILE(2, 40, 2, false, true)
  TILE(1, 48, 2, false, true)
  TILE(1, 80, 2, false, true)
  return 0;
}
