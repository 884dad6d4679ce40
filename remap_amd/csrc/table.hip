// table.hip — per-frame descriptor tables (gfx950).
//
// kpe_main appends keypoints in arbitrary order.  The reference emits them while walking
// x ascending (outer) and y ascending (inner) (kpe.hpp:201-205,289) and every map/vector
// insertion order downstream (kpr.hpp:121-124) follows from that scan order, so each
// frame's records are sorted by (x, y) here — keys are unique per frame, which makes the
// table bit-reproducible whatever the append order was — and written as rmx_desc rows
// with the region mask of kpe.hpp:157-192,235-277.
#include "common.h"

namespace rmx {

struct TableParams {
  const RawDesc* raw;
  const uint32_t* kcount;
  rmx_desc* desc;
  uint32_t* kout;  // clamped count per frame (min(k, kmax)); may alias kcount
  uint32_t* status;
  uint32_t n, kmax;
  Bands bands;
  uint32_t H;          // frame height (scan order = x * H + y)
  uint32_t rank_words;  // ceil(W * H / 32) if the rank bitmap fits the launch's LDS, else 0
  uint32_t wave_ok;     // frame <= 1024 x 1024 and kmax <= 2048: (x, y, index) fit a 32-bit sort key
  uint32_t* left;       // n flags: frame not done by the wave pass
  uint32_t* knext;      // the NEXT batch's append counters (the other of two buffers): cleared here, no memset launch
};

// In-LDS bitonic sort of n2 (power of two) 64-bit keys, ascending; blockDim.x threads.
__device__ inline void bitonic_sort_u64(uint64_t* a, uint32_t n2) {
  for (uint32_t k = 2; k <= n2; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < n2; i += blockDim.x) {
        uint32_t ixj = i ^ j;
        if (ixj > i) {
          uint64_t x = a[i], y = a[ixj];
          bool up = (i & k) == 0;
          if ((x > y) == up) {
            a[i] = y;
            a[ixj] = x;
          }
        }
      }
    }
  }
  __syncthreads();
}

__device__ __forceinline__ void write_row(const TableParams& p, rmx_desc* out, uint32_t i, const RawDesc& r) {
  uint32_t x = r.xy & 0xffffu, y = r.xy >> 16;
  uint32_t m = region_mask(p.bands, x, y);
  // 18-byte rows, 2-byte aligned: one unaligned 16-byte store and a 2-byte one (nine 2-byte stores otherwise)
  const uint4 v = make_uint4(r.code[0], r.code[1], r.code[2], (r.code[3] & 0xffu) | m << 8 | x << 16);
  const uint16_t yy = (uint16_t)y;
  uint8_t* o = reinterpret_cast<uint8_t*>(out + i);
  __builtin_memcpy(o, &v, 16);
  __builtin_memcpy(o + 16, &yy, 2);
}

// Three ways to the same table.
//  * Up to 256 keypoints and a frame of at most 1024 x 1024 (the common case): ONE WAVE per
//    frame sorts (x, y, index) packed into 32-bit keys in registers (wave_sort) — no LDS, no
//    workgroup barrier, four frames per workgroup.
//  * More keypoints: positions are unique, so a keypoint's row is its rank in scan order — one
//    bit per pixel in an LDS bitmap indexed x * H + y, popcount prefix per thread chunk, then
//    every record computes its own row.  O(pixels / 32 + k) instead of O(k log^2 k).
//  * Neither fits (huge frames): bitonic sort of 64-bit keys in LDS.
// The last two need the whole workgroup; they run after the wave pass for the frames it left.
__device__ __forceinline__ void kpe_table_block(const TableParams& p, uint64_t* keys, uint32_t f) {
  __shared__ uint32_t chunk_pre[256];
  __shared__ uint32_t wave_tot[4];
  uint32_t k = p.kcount[f];
  if (k > p.kmax) k = p.kmax;  // overflow already flagged by kpe_main
  const RawDesc* raw = p.raw + (size_t)f * p.kmax;
  rmx_desc* out = p.desc + (size_t)f * p.kmax;
  if (threadIdx.x == 0) p.kout[f] = k;
  if (k > 128 && p.rank_words) {
    uint32_t* bm = reinterpret_cast<uint32_t*>(keys);
    const uint32_t nb = p.rank_words, cw = (nb + 255) / 256;
    // the first RPT * 256 records stay in registers from here to their rows (all their loads in
    // flight together; one 4-byte and one 32-byte load per record in two loops otherwise)
    constexpr uint32_t RPT = 5;
    uint4 code[RPT];
    uint32_t xyr[RPT];
#pragma unroll
    for (uint32_t u = 0; u < RPT; ++u) {
      const uint32_t i = threadIdx.x + 256 * u;
      code[u] = make_uint4(0, 0, 0, 0);
      xyr[u] = 0;
      if (i < k) {
        code[u] = *reinterpret_cast<const uint4*>(raw[i].code);
        xyr[u] = raw[i].xy;
      }
    }
    for (uint32_t w = threadIdx.x; w < nb; w += 256) bm[w] = 0;  // (under the loads)
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < RPT; ++u)
      if (threadIdx.x + 256 * u < k) {
        const uint32_t b = (xyr[u] & 0xffffu) * p.H + (xyr[u] >> 16);
        atomicOr(&bm[b >> 5], 1u << (b & 31u));
      }
    for (uint32_t i = threadIdx.x + 256 * RPT; i < k; i += 256) {
      const uint32_t xy = raw[i].xy, b = (xy & 0xffffu) * p.H + (xy >> 16);
      atomicOr(&bm[b >> 5], 1u << (b & 31u));
    }
    __syncthreads();
    const uint32_t w0 = min(nb, threadIdx.x * cw), w1 = min(nb, w0 + cw);
    uint32_t cnt = 0;
    for (uint32_t w = w0; w < w1; ++w) cnt += __popc(bm[w]);
    // exclusive scan of the 256 chunk counts
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = (uint32_t)__shfl_up((int)incl, o);
      if (lane >= (uint32_t)o) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < wave; ++w) base += wave_tot[w];
    chunk_pre[threadIdx.x] = base + incl - cnt;
    __syncthreads();
    auto rank_of = [&](uint32_t xy) {
      const uint32_t b = (xy & 0xffffu) * p.H + (xy >> 16), w = b >> 5, c = w / cw;
      uint32_t rank = chunk_pre[c] + __popc(bm[w] & ((1u << (b & 31u)) - 1u));
      for (uint32_t v = c * cw; v < w; ++v) rank += __popc(bm[v]);
      return rank;
    };
#pragma unroll
    for (uint32_t u = 0; u < RPT; ++u)
      if (threadIdx.x + 256 * u < k) {
        RawDesc r;
        r.code[0] = code[u].x;
        r.code[1] = code[u].y;
        r.code[2] = code[u].z;
        r.code[3] = code[u].w;
        r.xy = xyr[u];
        write_row(p, out, rank_of(xyr[u]), r);
      }
    for (uint32_t i = threadIdx.x + 256 * RPT; i < k; i += 256) {
      const RawDesc& r = raw[i];
      write_row(p, out, rank_of(r.xy), r);
    }
    __syncthreads();  // LDS is reused by the workgroup's next frame
    return;
  }
  uint32_t n2 = 1;
  while (n2 < k) n2 <<= 1;
  for (uint32_t i = threadIdx.x; i < n2; i += blockDim.x) {
    uint64_t key = ~0ull;
    if (i < k) {
      uint32_t xy = raw[i].xy;
      key = (uint64_t)((xy << 16) | (xy >> 16)) << 32 | i;  // x major, y minor
    }
    keys[i] = key;
  }
  bitonic_sort_u64(keys, n2);
  for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) write_row(p, out, i, raw[(uint32_t)keys[i]]);
  __syncthreads();
}

// wave pass: four frames per workgroup, no LDS; flags the frames it leaves to the block pass
__global__ __launch_bounds__(256) void kpe_table(TableParams p) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t f = blockIdx.x * 4 + wave;
  if (f >= p.n) return;
  uint32_t k = p.kcount[f];
  if (lane == 0) p.knext[f] = 0;
  if (k > p.kmax) k = p.kmax;
  // keypoints of every 32nd frame, summed: the host's estimate of the stream's density (which finish the next
  // batch's extraction kernel uses, api.hip); the extraction kernel cleared the word
  if (lane == 0 && (f & 31u) == 0) atomicAdd(&p.status[15], k);
  const bool mine = p.wave_ok && k <= 256;
  if (lane == 0) p.left[f] = mine ? 0u : 1u;
  if (!mine) return;
  const RawDesc* raw = p.raw + (size_t)f * p.kmax;
  rmx_desc* out = p.desc + (size_t)f * p.kmax;
  uint32_t key[4];
#pragma unroll
  for (uint32_t sl = 0; sl < 4; ++sl) {
    const uint32_t i = sl * 64 + lane;
    key[sl] = 0xffffffffu;
    if (i < k) {
      const uint32_t xy = raw[i].xy;
      key[sl] = (xy & 0xffffu) << 21 | (xy >> 16) << 11 | i;  // x (10 bits), y (10), index (11)
    }
  }
  uint32_t n2 = 64;
  while (n2 < k) n2 <<= 1;
  wave_sort<4>(key, n2, lane);
#pragma unroll
  for (uint32_t sl = 0; sl < 4; ++sl) {
    const uint32_t j = sl * 64 + lane;
    if (j < k) write_row(p, out, j, raw[key[sl] & 0x7ffu]);
  }
  if (lane == 0) p.kout[f] = k;
}

// block pass: a fixed grid; workgroup b looks at the flags of frames b, b + G, b + 2G, ... side
// by side and walks only the flagged ones (an all-sparse batch costs the launch, nothing more)
__global__ __launch_bounds__(256) void kpe_table_rest(TableParams p) {
  extern __shared__ uint64_t keys[];  // the rank bitmap, or sort keys
  const uint32_t G = gridDim.x, b = blockIdx.x;
  const uint32_t K = b < p.n ? (p.n - b + G - 1) / G : 0;
  for (uint32_t base = 0; base < K; base += 256) {
    const uint32_t kk = base + threadIdx.x;
    if (!__syncthreads_or(kk < K && p.left[b + kk * G] != 0)) continue;
    const uint32_t end = min(K, base + 256);
    for (uint32_t q = base; q < end; ++q)
      if (p.left[b + q * G]) kpe_table_block(p, keys, b + q * G);  // uniform for the workgroup
  }
}

// Tables of any size (contexts with kmax above 16 384: the blended canvases of large fragments,
// fgs.hpp:80-89): the rank bitmap of kpe_table_block — a keypoint's row is the number of
// keypoints before it in scan order — kept in global memory, one workgroup of 1 024 threads per
// frame.  scratch: 2 * ceil(W * H / 32) dwords per frame (bits, running counts).
__global__ __launch_bounds__(1024) void kpe_table_big(TableParams p, uint32_t* scratch, uint32_t words) {
  __shared__ uint32_t s_base[1024];
  __shared__ uint32_t s_wave[16];
  const uint32_t f = blockIdx.x, tid = threadIdx.x;
  uint32_t k = p.kcount[f];
  if (tid == 0) p.knext[f] = 0;
  if (k > p.kmax) k = p.kmax;  // overflow already flagged
  const RawDesc* raw = p.raw + (size_t)f * p.kmax;
  rmx_desc* out = p.desc + (size_t)f * p.kmax;
  uint32_t* bm = scratch + (size_t)f * 2 * words;
  uint32_t* pre = bm + words;
  for (uint32_t w = tid; w < words; w += 1024) bm[w] = 0;
  __syncthreads();
  for (uint32_t i = tid; i < k; i += 1024) {
    const uint32_t xy = raw[i].xy, b = (xy & 0xffffu) * p.H + (xy >> 16);
    atomicOr(&bm[b >> 5], 1u << (b & 31u));
  }
  __syncthreads();
  const uint32_t per = (words + 1023) / 1024;
  const uint32_t w0 = min(words, tid * per), w1 = min(words, w0 + per);
  uint32_t cnt = 0;
  for (uint32_t w = w0; w < w1; ++w) {
    pre[w] = cnt;  // set bits before word w inside the thread's chunk
    cnt += __popc(bm[w]);
  }
  const uint32_t lane = tid & 63u, wave = tid >> 6;
  uint32_t incl = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)incl, o);
    if (lane >= (uint32_t)o) incl += t;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (uint32_t w = 0; w < wave; ++w) base += s_wave[w];
  s_base[tid] = base + incl - cnt;
  __syncthreads();
  for (uint32_t i = tid; i < k; i += 1024) {
    const RawDesc& r = raw[i];
    const uint32_t b = (r.xy & 0xffffu) * p.H + (r.xy >> 16), w = b >> 5;
    write_row(p, out, s_base[w / per] + pre[w] + __popc(bm[w] & ((1u << (b & 31u)) - 1u)), r);
  }
  if (tid == 0) p.kout[f] = k;
}

}  // namespace rmx
